// Implicit-GEMM convolution on the gfx950 matrix cores, fp32 in / fp32 accumulate.
//
// Replaces the cuDNN conv2d/conv3d + batch_norm + activation (+ residual, + GRU gate arithmetic) call
// chains of the reference's BEV stack (fiery/layers/convolutions.py:9-168, fiery/layers/temporal.py:10-281,
// fiery/models/decoder.py:53-91).  One kernel covers 1x1, 3x3 (stride 1/2), 7x7 stride 2 and the causal
// (kT,3,3) temporal convolutions: the GEMM is  out[pixel][cout] = sum_k A[pixel][k] * W[k][cout]  with
// k = (tap, input channel), A gathered on the fly from pixel-major (NHWC) activations.
//
// Mapping to CDNA4:
//   * v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (the 1e-4 parity budget rules out
//     bf16 for this configuration); 64 cycles per instruction per SIMD, so the matrix pipe is the
//     bound and everything else is sized to stay out of its way.
//   * workgroup = 4 wavefronts, tile = 128 pixels x BN couts (BN = 64: 2x2 wavefronts of 64x32;
//     BN = 32: 4x1 wavefronts of 32x32), K advanced 32 at a time through double-buffered LDS, one
//     barrier per step; global loads for step k+1 are in flight while step k runs on the MFMAs.
//   * A tile is stored [pixel][32 k] with the 16-byte slot index XOR-ed by (pixel>>1)&7, so the
//     ds_read_b128 of 32 consecutive pixels at one k-slot is bank-conflict free; each b128 feeds four
//     MFMA k-steps (the K order inside a step is permuted identically for A and W).
//   * the weight tile is pre-packed on the host side in exactly its LDS image ([k/4][cout][k%4], so a
//     lane's four k of an MFMA group are one ds_read_b128): one contiguous 4/8/16 KiB copy per step.
//   * input channels are a virtual concat of two tensors (GRU [x, h], temporal-block path concat), and
//     the epilogue applies bias / folded BatchNorm / activation / residual / GRU gate math in registers.
//
// This header holds the kernel template; every tile shape is instantiated in a translation unit of its own
// (conv_tile_<BM>x<BN>.hip) so that the five shapes compile side by side, and conv_igemm.hip holds the C ABI.
#pragma once
#include "common.h"
#include <fiery_gfx950.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fiery {

typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef FIERY_CONV_EARLY_LOADS
#define FIERY_CONV_EARLY_LOADS 1
#endif
constexpr int BK = 32;    // k per LDS stage (4 units of 8 input channels)
#ifndef FIERY_CONV_EPILOGUE_PRIO
#define FIERY_CONV_EPILOGUE_PRIO 0     // wave priority of the epilogue (0 = unchanged); see the end of the K loop
#endif
// A fourth workgroup per CU for the scalar-addressed 128 x 32 kernel (A/B switch): measured in round 3 after its registers
// were trimmed to fit - faster for the plain and the chained launch (106 vs 112, 131 vs 138 us), slower for the Bottleneck
// tail with its third stage, the launch the step issues most (162 vs 153 us): off.  Measured again after that epilogue
// stopped spilling at 128 registers (residual rows in two halves): 107 vs 115, 134 vs 139, 149 vs 158 - and 162 vs 155 for the
// third-stage launch, 286.5 vs 289.1 samples/s (profiles/r3_tail_four_per_cu_ab.txt): still off.
#ifndef FIERY_TAIL_FOUR_PER_CU
#define FIERY_TAIL_FOUR_PER_CU 0
#endif
struct SrcP {
    const float* ptr;
    int ld, units;
    long long bstride, tstride;
    int ext_bytes;        // bytes from ptr to the end of the last row this launch may read (scalar-addressed loop: the
                          // buffer descriptor's size, so that the hardware returns zeros for anything beyond it)
};
struct TensP {
    float* ptr;
    int ld;
    long long istride;
};
struct HeadsP {
    const float* w;
    const float* bias;
    int n_out;
    int group[FIERY_MAX_HEAD_OUTPUTS], sigmoid[FIERY_MAX_HEAD_OUTPUTS];
    float* out[FIERY_MAX_HEAD_OUTPUTS];
    long long istride[FIERY_MAX_HEAD_OUTPUTS];
};
struct ConvP {
    SrcP src[2];
    int Hin, Win, Hout, Wout, n_img, Tout, tout0, tinadd;
    int kT, kH, kW, stride, padH, padW;
    const float* w;
    int cout_pad, k_chunks, n_units, cin_units;
    const float* scale;
    const float* shift;
    const float* img_bias;
    int act, epi, res_pre;
    TensP res, out, out2, aux0, aux1;
    int cout_store;
    long long M;
    int tiles_m;          // pixel tiles of the launch (the grid's x extent may be smaller: persistent workgroups, see k_conv_igemm)
    // exact division of a 31-bit pixel index by Hout Wout, Wout and Tout as multiply + shift (set by the host, see
    // conv_magic): the kernel's set-up divides a dozen times per thread, and a run-time divisor costs ~25 vector
    // instructions per division on this ISA
    unsigned mg_hw, mg_w, mg_t;
    int sh_hw, sh_w, sh_t;
    const float* w2;      // chained 1x1 (BN = 32 only)
    const float* scale2;
    const float* shift2;
    int act2;
    int vec_epilogue;     // bit 0: destinations / residual / bias rows are 16-byte addressable; bit 1: and dense (see store_rows);
                          // bit 2: every tensor spans < 2 GB from its base at a non-negative image stride (conv_winograd.hip)
    int bias_border;      // img_bias holds nine rows per image, chosen by the output pixel's border class
    HeadsP heads;         // FIERY_EPI_HEADS
    // stream-K launches (SK kernels): output tiles of the launch (pixel tiles x cout tiles), partial tiles' workspace
    // (two slots of BM x BN floats per workgroup), one ticket counter per output tile (zero between launches)
    int sk_tiles;
    float* sk_ws;
    int* sk_cnt;
    // Winograd launches: exact division of a block index by the blocks per image / per block row (multiply + shift, as mg_hw)
    unsigned wmg_img, wmg_tw;
    int wsh_img, wsh_tw;
};

// kernel variants of one tile shape
enum ConvVariant {
    kConvGeneric = 0,      // per-lane addressed K loop: any channel layout
    kConvAligned = 1,      // scalar-addressed K loop (every tap holds whole 32-channel stages of one source)
    kConvSmallCin = 2,     // fewer than four 8-channel units per tap
    kConvClock = 3,        // tuning builds (FIERY_CONV_TUNING): clock/phase probe, generic loop
    kConvClockAligned = 4, //                                      clock/phase probe, scalar-addressed loop
    kConvPrio = 5,         //                                      wave-priority experiment
};
// one launcher per tile shape, each in its own translation unit; returns false when the variant is not built
bool conv_launch_128x32(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
bool conv_launch_128x64(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
bool conv_launch_128x128(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
bool conv_launch_128x128_rest(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
bool conv_launch_64x64(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
bool conv_launch_64x128(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk);
// bf16 matrix-core form of the scalar-addressed kernel (weights packed by fiery_conv_pack_weights_bf16 in p.w); returns
// false when (bm, bn) has no such kernel
bool conv_launch_bf16(const ConvP& p, int bm, int bn, dim3 grid, hipStream_t stream, bool halo = false);
// split form of the scalar-addressed kernel (conv_tile_split.hip): 128-pixel tiles, bn = 32 (the chained tails too) or 64
bool conv_launch_split(const ConvP& p, int bn, dim3 grid, hipStream_t stream);
// weight packers' argument: padded channel position -> logical input channel, -1 = padding (2 KB of kernel arguments)
constexpr int kMaxPackUnits = 128;      // 8-channel input units per convolution: 1024 channels (the trunk's 960-wide projections)
struct ChanInverse {
    short ci[kMaxPackUnits * 8];
};
// Winograd F(2x2, 3x3) form (conv_winograd.hip): 3 x 3 / stride 1 / 'same' layers, 64-cout tiles, 16-channel stages
size_t conv_winograd_packed_floats(int cout, int cin_units);
int conv_winograd_pack(const float* w, int cout, int cin_total, const ChanInverse& inv, int cin_units, float* packed, hipStream_t stream);
bool conv_launch_winograd(const ConvP& p, hipStream_t stream);
// its split form (conv_winograd_split.hip): bf16 matrix cores, operands as three bf16 terms, fp32 accuracy
size_t conv_winograd_split_packed_floats(int cout, int cin_units);
int conv_winograd_split_pack(const float* w, int cout, int cin_total, const ChanInverse& inv, int cin_units, float* packed, hipStream_t stream);
bool conv_launch_winograd_split(const ConvP& p, hipStream_t stream);
// stream-K form of the scalar-addressed fp32 kernel on 128-pixel tiles (bn = 64 or 128); returns false when bn has none
bool conv_launch_stream_k(const ConvP& p, int bn, dim3 grid, hipStream_t stream);
// workgroups the stream-K kernel of cout tile width bn keeps resident per CU
int conv_stream_k_per_cu(int bn);
// fp32 halo loop (3 x 3 / stride 1 layers on 64-pixel tiles; bn = 64 or 128)
bool conv_launch_f32_halo(const ConvP& p, int bn, dim3 grid, hipStream_t stream);

#ifdef FIERY_CONV_KERNEL_TU
namespace {

// tuning aid (FIERY_CONV_CLKPROBE): {sum of shader cycles, sum of 100 MHz ticks} over the K loops of sampled workgroups;
// a __device__ global rather than a ConvP member so that the production kernel's argument block (and with it its
// register allocation, which is touchy) is exactly what it is without the probe
__device__ unsigned long long* g_clk_probe = nullptr;
// what a tap outside the image (zero padding) or past the end of K reads
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};   // (not const: it must live in the global address space like the sources, or the select makes the loads flat)

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }
// The GRU gates' sigmoid in the row epilogues: v_exp_f32 and v_rcp_f32 (1 ulp each) - five instructions where the correctly
// rounded division and libm's expf take about eighteen; a gate launch's epilogue applies it to 32 values per lane, and epilogue
// instructions are the expensive kind.  1e-7 on a gate in [0, 1].
__device__ __forceinline__ float sigmoid_gate(float v) {
#if defined(FIERY_SIGMOID_LIBM) && FIERY_SIGMOID_LIBM        // A-B builds: libm's expf and the correctly rounded division (parity experiments)
    return sigmoidf(v);
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
#endif
}
// The swish epilogue's sigmoid (the image trunk's 1x1 expansions: K = 24 .. 160 channels - one to five stages of MFMAs - then 48
// activations per lane: those launches are bound by the epilogue's vector instructions, 2.9x off their HBM bound with libm's expf
// and the correctly rounded division).  FIERY_SWISH_FAST=0: libm (A-B builds).
#ifndef FIERY_SWISH_FAST
#define FIERY_SWISH_FAST 1
#endif
__device__ __forceinline__ float sigmoid_swish(float v) {
#if FIERY_SWISH_FAST
    return sigmoid_gate(v);
#else
    return sigmoidf(v);
#endif
}
// g / d for 0 <= g < 2^31 with (m, s) = conv_magic(d): exact (Granlund-Montgomery, round-up form)
__device__ __forceinline__ int fast_div(int g, unsigned m, int s) {
    return static_cast<int>((static_cast<unsigned long long>(static_cast<unsigned>(g)) * m) >> s);
}

// BM output pixels x BN couts per workgroup: (128, 32|64|128) and (64, 64|128).  The 64-pixel tiles exist for
// launches whose 128-pixel tile count would leave a badly filled last wave of workgroups.
// Workgroups of one tile shape that fit a CU, as its LDS (two A stages + two W stages) allows, capped where the tile's
// registers would not follow: one wavefront per workgroup and SIMD, so this is also the waves-per-SIMD target that the
// register allocation is held to (without it the compiler aims one notch too high for the 64 x 128 tile and spills).
// (halo loop, 3 x 3: three runs of bm + 2 pixels + one zero entry; an entry is 32 channels - 64 bytes of bf16 or 128 of fp32 -
// plus 16 bytes that keep ds_read_b128 conflict-free: 20 / 36 floats)
constexpr int HALO_RUN_EXTRA = 3;
constexpr int halo_pitch(bool bf16) { return bf16 ? 20 : 36; }
// (chained tails, round 5: each wavefront has an exchange tile of its own - 32 pixel rows of 64 couts, 68 floats apart - in
// which accumulators in the lane-per-pixel layout become full rows and back; no staging tile)
constexpr int CHAIN_PITCH = 68;
constexpr int conv_smem_floats(int bm, int bn, bool bf16, bool halo = false, bool chain = false, bool split = false) {
    const int full = 2 * bm * BK + 2 * BK * bn;
    if (!halo && !bf16) return full;
    // (split form: three bf16 images of either operand)
    // (... of the weights always; of the A tile where several wavefronts read an element - 64-wide cout tiles; on the 32-wide
    // tiles every A element has ONE reader, the tile stays fp32 in LDS and is split by that reader: a third less LDS, three
    // workgroups per CU instead of two)
    const int stages = halo ? 3 * (bm + HALO_RUN_EXTRA) * halo_pitch(bf16)
                            : split ? (bn == 32 ? 2 * bm * BK + 3 * BK * bn : 3 * (full / 2)) : full / 2;
    const int epilogue = chain ? 4 * 32 * CHAIN_PITCH : bm * bn + (bm == 64 && bn == 128 ? 256 : 0);       // staging tile (+ the decoder heads' 1x1 rows)
    return stages > epilogue ? stages : epilogue;
}
#ifndef FIERY_HALO_F32_WAVES
#define FIERY_HALO_F32_WAVES 3    // fp32 halo loop (opt-in): 4 = 128 registers, 1,024 workgroup slots - measured equal (profiles/r3_conv_halo_f32_waves_ab.txt)
#endif
#ifndef FIERY_BF16_WAVES
#define FIERY_BF16_WAVES 4        // waves per SIMD the bf16 form's register allocation is held to (128 registers)
#endif
constexpr int conv_waves_per_simd(int bm, int bn, bool aligned = false, bool bf16 = false, bool halo = false, bool chain = false, bool split = false) {
    const int by_lds = 163840 / (conv_smem_floats(bm, bn, bf16, halo, chain, split) * 4);
    if (split) return by_lds < (bn == 32 ? 3 : 2) ? by_lds : (bn == 32 ? 3 : 2);      // (three operand images per MFMA block in registers)
    int cap = ((bm == 64 && bn == 64) || (FIERY_TAIL_FOUR_PER_CU && bm == 128 && bn == 32 && aligned)) ? 4 : 3;
    if (bf16 && !(bm == 128 && bn == 32)) cap = FIERY_BF16_WAVES;         // (128 x 32: its chained epilogue needs 150)
    if (halo && !bf16) cap = FIERY_HALO_F32_WAVES;
    if (halo && bm == 128) cap = bn == 128 ? 2 : 3;                        // (13 halo elements per thread in flight)
    return by_lds < cap ? by_lds : cap;
}

// BF16: the matrix cores run v_mfma_f32_32x32x16_bf16 - sixteen k per instruction instead of two, an eighth of the
// matrix-pipe time per stage.  Activations stay fp32 in HBM (the gather and every epilogue are the fp32 kernel's); a
// thread rounds the four k it gathered to bf16 (round to nearest even, v_cvt_pk_bf16_f32) as it writes them to LDS,
// so the A tile there is [pixel][32 k] bf16 - 64-byte rows of four 16-byte slots, slot s of row r stored at
// s ^ ((r >> 2) & 3) (the 16 lanes of a ds_read_b128 group then touch 16 different slots of the 256-byte bank row) -
// and a lane's operand of an MFMA is ONE 16-byte read.  (Until round 3 the tile stayed fp32 in LDS and was rounded at
// operand-read time: twice the LDS bytes written, twice the reads, and every value converted once per wavefront that
// used it - with bf16's short MFMAs the loop was bound by LDS traffic, 3 KB per MFMA against the 1 KB per MFMA the LDS
// can deliver at full matrix rate.)  The weights arrive already rounded and packed [k / 8][cout][k % 8]; products are
// exact and accumulate in fp32.  Scalar-addressed loop only.
// One output tile (pixel tile `bid_x` of `nblk_x` in dispatch order, cout tile blockIdx.y); the kernel below walks a
// workgroup through its tiles.
// (SK - stream-K, round 5: the workgroup multiplies chunks [sk_kb, sk_ke) of output tile sk_tile; sk_j = its place in the
// launch's even split of all (tile, chunk) units; see k_conv_igemm)
// SPLIT (round 6; BF16 kernels only): fp32 ACCURACY on the bf16 matrix cores - every operand is the sum of three bf16 terms
// (x = x1 + x2 + x3 exactly) and a product the six partial products of weight >= 2^-24, smallest first (see conv_winograd.hip,
// "SPLIT form", and tools/probe/split_bf16_probe.hip: not less accurate than the fp32 matrix instruction).  The A tile is
// split as it is written to LDS (three bf16 images), the weights arrive split (fiery_conv_pack_weights_split: per stage
// [term][k / 8][cout][k % 8]); twelve 8-pass MFMAs per 32 x 32 x 32 block instead of sixteen 16-pass ones.
template <int BM, int BN, bool CLK, int PRIO, bool SMALLCIN, bool ALIGNED, bool BF16, bool HALO, bool CHAIN, bool SK = false, bool SPLIT = false>
__device__ __forceinline__ void conv_tile(const ConvP& p, const int bid_x, const int nblk_x, const int sk_tile = 0, const int sk_kb = 0,
                                          const int sk_ke = 0, const int sk_j = 0, const int sk_nwg = 0) {
    static_assert(!CHAIN || (BM == 128 && BN == 32 && !HALO), "the chained tails run on the 128 x 32 tile");
    static_assert(!SK || (ALIGNED && !BF16 && !HALO && !CHAIN && !CLK && !SMALLCIN), "stream-K: the scalar-addressed fp32 loop");
    static_assert(!BF16 || (ALIGNED && !SMALLCIN && !CLK), "the bf16 form exists for the scalar-addressed loop");
    static_assert(!SPLIT || (BF16 && !HALO && BN <= 64), "the split form: the scalar-addressed bf16 loop, cout tiles of 32 or 64");
    constexpr int NS = SPLIT ? 3 : 1;                 // bf16 terms per operand
    // SPLIT_READ: the A tile stays fp32 in LDS (the fp32 kernel's image) and the wavefront that multiplies a row splits it as it
    // reads it - on the 32-wide cout tiles (one wavefront column) nobody else reads that row, so the split costs the same vector
    // instructions as at the writing side and the tile takes 16 KB per stage instead of 24
    constexpr bool SPLIT_READ = SPLIT && BN == 32;
    static_assert(!HALO || (ALIGNED && !SMALLCIN && !CLK && BN >= 64 && (BM == 64 || BF16)), "the halo loop: 64-pixel tiles (bf16: 128 too), 64 couts or more");
    unsigned long long clk_entry = 0;
    // (tuning builds: slot 7 of the probe block may hold the address of a timeline buffer - eight 64-bit words per tile: the
    // 100 MHz wall clock at entry / K loop start / K loop end / end of the tile, and the hardware id of wave 0's SIMD)
    unsigned long long* clk_trace = nullptr;
    if constexpr (CLK) {
        clk_entry = clock64();
        if (threadIdx.x == 0 && g_clk_probe && g_clk_probe[7]) {
            clk_trace = reinterpret_cast<unsigned long long*>(g_clk_probe[7]) + 8ull * (bid_x + static_cast<unsigned>(blockIdx.y) * nblk_x);
            clk_trace[0] = wall_clock64();
            clk_trace[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
            clk_trace[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
        }
    }
    if constexpr (PRIO == 1) {
        // experiment: workgroups that share a CU (dispatch order puts b and b + 256 on one CU first) get different wave
        // priorities, so that they do not march through their MFMA and load/store phases in lockstep
        switch ((bid_x >> 8) % 3) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: break;
        }
    }
    constexpr int NA = BM / 32;            // A-gather loads (16 B each) per thread and stage
    // (halo loop: a wavefront fetches its own weights, so with 128 couts each wavefront takes 32 of them and all 64 pixels)
    constexpr int WN = HALO ? (BN >= 128 ? 4 : 2) : (BN >= 64 ? 2 : 1);   // wavefronts along couts
    constexpr int WM = 4 / WN;             // wavefronts along pixels
    constexpr int MT = BM / (32 * WM);     // 32-pixel MFMA tiles per wavefront   (BN=32: 1, else 2)
    constexpr int NT = BN / (32 * WN);     // 32-cout MFMA tiles per wavefront    (BN=128: 2, else 1)
    // PIXLANE (the chained tails' kernel - CHAIN - on the 128 x 32 tile, round 5): the MFMAs take the WEIGHTS as their A operand and the pixels as B, so the
    // accumulator block is the tile TRANSPOSED - lane l holds pixel l & 31 of its wavefront's 32, register r holds cout
    // c(r, hi) = 8 (r / 4) + 4 hi + r % 4.  That is (a) the operand layout of a following 32-k MFMA step (lane = pixel, lane
    // half = k), so the Bottleneck tails' chained 1 x 1 products read their input straight from the accumulator registers -
    // no LDS round trip, no barrier, wavefronts finish on their own - and (b) four consecutive couts in four consecutive
    // registers, so a lane finishes 16-byte pieces of its own pixel row without a staging tile (tools/probe/
    // chained_gemm_swapped.hip, profiles/r5_chained_gemm_swapped_gpu.txt: the layout algebra on real MFMA lanes).
    constexpr bool PIXLANE = CHAIN;
    constexpr int W_BYTES = BF16 ? 2 * NS : 4;                 // bytes per packed weight
    constexpr int BLOADS = BF16 ? (SPLIT ? (BN >= 64 ? 3 : 2) : (BN >= 64 ? BN / 64 : 1)) : (BK * BN / 4) / 256;      // 16-byte W loads per thread and stage

    // one LDS block: two A stages, two W stages; the epilogue reuses it as a BM x BN staging tile.  The bf16 form's stages
    // are half the size (both operands are bf16 there), so its block is as large as the epilogue needs and no larger.
    constexpr bool HALF_STAGES = BF16 && !HALO;
    constexpr int A_PLANE = BM * (BK / 2);                                // floats of one bf16 image of the A tile
    constexpr int W_PLANE = BK * BN / 2;
    constexpr int A_STAGE = SPLIT_READ ? BM * BK : HALF_STAGES ? NS * A_PLANE : BM * BK;         // floats between the two A stages
    constexpr int W_STAGE = HALF_STAGES ? NS * W_PLANE : BK * BN;
    constexpr int W_BASE = 2 * A_STAGE;
    constexpr int SMEM_FLOATS = conv_smem_floats(BM, BN, BF16, HALO, CHAIN, SPLIT);
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    static_assert((HALO || W_BASE + 2 * W_STAGE <= SMEM_FLOATS) && (CHAIN ? 4 * 32 * CHAIN_PITCH : BM * BN + (BM == 64 && BN == 128 ? 256 : 0)) <= SMEM_FLOATS,
                  "stages, staging tile (chained tails: the wavefronts' exchange tiles) and the heads' 1x1 rows must fit");

    // (taken afresh for every tile of the persistent loop: hoisted out of it, everything derived from the thread index -
    // a few dozen registers that the set-up needs and the K loop does not - would stay live across the whole tile)
#if FIERY_CONV_EPILOGUE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const int sk_ntiles = SK ? p.cout_pad / BN : 1;
    int tid_ = threadIdx.x, tile_n_ = SK ? sk_tile % sk_ntiles : blockIdx.y;
    opaque_v(tid_);
    opaque_s(tile_n_);
    const int tid = tid_;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv % WM, wn = wv / WM;
    const int m = lane & 31, hi = lane >> 5;
    const int tile_n = tile_n_;
    // XCD-aware tile order: workgroup b runs on XCD b % 8, so give every XCD a contiguous run of pixel
    // tiles - neighbouring tiles share their 3x3 halo rows through that XCD's L2 instead of re-fetching them
    int tile_m;
    if constexpr (SK) {
        tile_m = sk_tile / sk_ntiles;                 // (the split itself is XCD-aware: k_conv_igemm)
    } else {
        const int nblk = nblk_x, bid = bid_x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int pix0 = tile_m * BM;             // M < 2^31 is checked on the host: 32-bit pixel arithmetic throughout
    const int HWout = p.Hout * p.Wout;
    const int M = static_cast<int>(p.M);

    // ---- this thread's share of the A gather: one 16-byte slot of 4 pixels per stage ----------------
    // Everything that does not change from stage to stage is computed once here, as 32-bit element offsets
    // (the host checks that every source spans < 2^31 floats); a stage then costs one add and a bounds
    // predicate per load, and the (tap, channel-unit) of the thread's slot advances incrementally.
    const int f4 = tid & 7;                // logical 16-byte slot inside the 32-k row
    const int prow = tid >> 3;             // 0..31
    int py0[NA], px0[NA], ptmin[NA], poff0[NA], poff1[NA];
    bool pvalid[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int gp = pix0 + prow + 32 * j;
        pvalid[j] = gp < M;
        const int g = pvalid[j] ? gp : 0;
        const int o = fast_div(g, p.mg_hw, p.sh_hw);
        const int rem = g - o * HWout;
        const int y = fast_div(rem, p.mg_w, p.sh_w), x = rem - y * p.Wout;
        const int b = fast_div(o, p.mg_t, p.sh_t), tl = o - b * p.Tout;
        py0[j] = y * p.stride - p.padH;
        px0[j] = x * p.stride - p.padW;
        ptmin[j] = tl + p.tout0 - (p.kT - 1);                       // absolute time of the dt = 0 tap
        const int pos = py0[j] * p.Win + px0[j];
        poff0[j] = static_cast<int>(b * p.src[0].bstride + (tl + p.tinadd - (p.kT - 1)) * p.src[0].tstride) + pos * p.src[0].ld;
        poff1[j] = static_cast<int>(b * p.src[1].bstride + (tl + p.tinadd - (p.kT - 1)) * p.src[1].tstride) + pos * p.src[1].ld;
    }
    const int taps = p.kT * p.kH * p.kW;
    const int c_Win = p.Win, c_Hin = p.Hin, c_cin_units = p.cin_units, c_kW = p.kW, c_kH = p.kH, c_k_chunks = p.k_chunks;
    // the two sources' fields as scalars (selecting between p.src[0].x and p.src[1].x directly turns into a dynamic
    // index into the argument block, which then has to live in scratch)
    const float* const src0_ptr = p.src[0].ptr;
    const float* const src1_ptr = p.src[1].ptr;
    const int src0_ld = p.src[0].ld, src1_ld = p.src[1].ld, src0_units = p.src[0].units;
    const int src0_ts = static_cast<int>(p.src[0].tstride), src1_ts = static_cast<int>(p.src[1].tstride);
    const int src0_ext = p.src[0].ext_bytes, src1_ext = p.src[1].ext_bytes;

    float4 areg0, areg1, areg2, areg3;          // named, like breg*: an indexed array that lives across iterations ends up in scratch
    areg0 = areg1 = areg2 = areg3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 breg0 = make_float4(0.f, 0.f, 0.f, 0.f), breg1 = breg0, breg2 = breg0, breg3 = breg0;

    // ---- the side work of a stage, cut into pieces that go between two MFMAs ------------------------------------------
    // Measured on gfx950 (tools/probe/mfma_valu_probe.hip): a VALU instruction issued between two fp32 MFMAs does NOT run
    // in their shadow - it takes ~3-6 cycles of the same pipe - so the K loop is built to need as few vector ALU
    // instructions as possible, not merely to hide them:
    //   * ALIGNED (every tap holds a whole number of stages from one source - all the big layers): the (tap, channel
    //     group) of a stage is the same for the whole workgroup, so it lives in SGPRs and is advanced by the scalar unit;
    //     a thread's pixel offsets never change; the tap's offset goes into the buffer load's scalar offset; zero padding
    //     is the buffer's out-of-range rule (an invalid tap gets an offset beyond the descriptor's size and reads 0).
    //     Per 16-byte load that leaves one mask test and one select on the vector ALU.
    //   * otherwise the thread's own (tap, unit) advances on the vector ALU, and an invalid tap reads the zero page.
    // Both ways the stage body is straight-line code, and the LDS addresses of a stage are thread constants plus
    // immediates because the two buffers are two copies of the body.
    constexpr int N_PIECES = 2 * (NA + BLOADS) + 4;
    //   generic path state
    int u_cc = 0, u_tap = 0, u_dt = 0, u_dy = 0, u_dx = 0;
    const float* wnext = p.w + static_cast<long long>(tile_n) * p.k_chunks * (BK * BN) + tid * 4;
    const float* const wfirst = wnext;
    int ld_stage = 0;
    const float* const zero_page = g_zero_page;
    bool ld_valid = false, ld_second = false;
    const float* ld_base = nullptr;
    int ld_tap_off = 0;
    //   aligned path state
    int voff0[NA], voff1[NA];                   // this thread's pixels in the two sources, bytes, >= 0
    unsigned vmask[NA];                         // rows | columns << 8 | frames << 16 of the kernel window that lie inside the
                                                // image for pixel j (0 when the pixel does not exist): tap (dt, dy, dx) is
                                                // inside iff its three bits are set
    int s_tap = 0, s_g = 0, s_dt = 0, s_dy = 0, s_dx = 0;      // wave-uniform
    int s_off = 0;
    unsigned s_sel = 0;                                        // the running tap's three bits (see vmask)
    bool s_second = false;
    __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src0_ptr), 0, 0, 0x00020000);
    const int s_groups = c_cin_units >> 2, s_groups0 = src0_units >> 2;
    if constexpr (SK) {
        // the pipeline's state is a function of the chunk it starts from: chunk = tap * groups + g, tap = (dt kH + dy) kW + dx
        ld_stage = sk_kb;
        s_tap = sk_kb / s_groups;
        s_g = sk_kb - s_tap * s_groups;
        const int khw = c_kH * c_kW;
        s_dt = s_tap / khw;
        const int r = s_tap - s_dt * khw;
        s_dy = r / c_kW;
        s_dx = r - s_dy * c_kW;
    }
    if constexpr (ALIGNED) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int gp = pix0 + prow + 32 * j;
            const bool pv = gp < M;
            const int g = pv ? gp : 0;
            const int o = fast_div(g, p.mg_hw, p.sh_hw);
            const int rem = g - o * HWout;
            const int y = fast_div(rem, p.mg_w, p.sh_w), x = rem - y * p.Wout;
            const int b = fast_div(o, p.mg_t, p.sh_t), tl = o - b * p.Tout;
            const int sp = (y * p.stride) * c_Win + x * p.stride;
            const int kofs = (f4 >> 1) * 8 + (f4 & 1) * 4;
            // (32-bit throughout: the host takes this loop only for sources that span less than 2^29 floats)
            voff0[j] = 4 * (b * static_cast<int>(p.src[0].bstride) + (tl + p.tinadd) * src0_ts + sp * src0_ld + kofs);
            voff1[j] = 4 * (b * static_cast<int>(p.src[1].bstride) + (tl + p.tinadd) * src1_ts + sp * src1_ld + kofs);
            // The window's valid rows / columns / frames are contiguous ranges, so each mask is two shifts - no loop over
            // taps.  (The prologue used to build a 64-bit tap mask with loops over kH, kW, kT and all taps: ~1,500 vector
            // instructions per workgroup, 19.8 k cycles measured - as long as the whole K loop of a 3x3 32 -> 32 layer,
            // and every one of them taken from the matrix pipe of the wavefronts sharing the SIMD.)
            auto range_bits = [](int lo, int hi) {             // bits [lo, hi) of a window of at most 8
                lo = lo < 0 ? 0 : lo;
                return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
            };
            const int y_in = y * p.stride - p.padH, x_in = x * p.stride - p.padW;      // window origin in the input
            const unsigned rowm = range_bits(-y_in, c_kH < c_Hin - y_in ? c_kH : c_Hin - y_in);
            const unsigned colm = range_bits(-x_in, c_kW < c_Win - x_in ? c_kW : c_Win - x_in);
            const unsigned tm = range_bits(p.kT - 1 - tl - p.tout0, p.kT);
            vmask[j] = pv ? (rowm | (colm << 8) | (tm << 16)) : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            voff0[j] = voff1[j] = 0;
            vmask[j] = 0;
        }
        // running decode of this thread's unit: u = stage*4 + (f4 >> 1) = tap * cin_units + cc
        const int u = f4 >> 1;
        u_tap = u / p.cin_units;
        u_cc = u - u_tap * p.cin_units;
        const int khw = p.kH * p.kW;
        u_dt = u_tap / khw;
        const int r = u_tap - u_dt * khw;
        u_dy = r / p.kW;
        u_dx = r - u_dy * p.kW;
    }
    // weights: one descriptor for this cout tile's packed image; thread t reads bytes [16 t, 16 t + 16) of every 4 KiB
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + static_cast<long long>(tile_n) * p.k_chunks * (BK * BN * W_BYTES)), 0,
        p.k_chunks * (BK * BN * W_BYTES), 0x00020000);             // exactly this cout tile's packed image
    // (bf16, BN = 32: a stage's weights are 2 KiB, half the threads have nothing to fetch and point past the descriptor)
    // (split, BN = 32: 6 KiB - every thread fetches a first piece, the lower half of the threads a second)
    const int w_voff = (BF16 && !SPLIT && BN == 32 && tid >= 128) ? static_cast<int>(0x80000000u) : tid * 16;
    const int w_voff_last = (SPLIT && BN == 32 && tid >= 128) ? static_cast<int>(0x80000000u) : w_voff;
    int w_soff = 0;
    auto to_float4 = [](auto raw) {
        float4 f;
        __builtin_memcpy(&f, &raw, 16);
        return f;
    };

    // (the scalar unit issues in order with the MFMAs: the stage's scalar bookkeeping is cut in three so that no single
    // gap between two MFMAs has to take all of it)
    int s_ld = 0, s_ts = 0, s_ext = 0;
    const float* s_base = nullptr;
    auto load_setup = [&](int part) {
        if constexpr (ALIGNED) {
            if (part == 0) {
                s_second = s_g >= s_groups0;
                s_base = s_second ? src1_ptr : src0_ptr;
                s_ld = s_second ? src1_ld : src0_ld;
                s_ts = s_second ? src1_ts : src0_ts;
                s_ext = s_second ? src1_ext : src0_ext;
                w_soff = (ld_stage < c_k_chunks ? ld_stage : c_k_chunks - 1) * (BK * BN * W_BYTES);      // past the end: repeat
            } else if (part == 1) {
                // the descriptor's base sits (kT-1) frames and (padH, padW) pixels before the source, so that the
                // tap's offset below is never negative; it ends with the last row the launch may read: the range check
                // covers vector + scalar offset (tools/probe/buffer_oob_probe.hip), so nothing beyond is ever touched
                const int lead = (p.kT - 1) * s_ts + (p.padH * c_Win + p.padW) * s_ld;
                s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s_base - lead), 0, 4 * lead + s_ext, 0x00020000);
            } else {
                s_off = 4 * (s_dt * s_ts + (s_dy * c_Win + s_dx) * s_ld + (s_g - (s_second ? s_groups0 : 0)) * 32);
                // (past the end of K the frame counter has run out of the window: bit 16 + kT is never set in a mask)
                s_sel = (1u << s_dy) | (1u << (8 + s_dx)) | (1u << (16 + (s_dt < 15 ? s_dt : 15)));
            }
        } else if (part == 0) {
            ld_valid = u_tap < taps;
            ld_second = u_cc >= src0_units;
            ld_base = ld_second ? src1_ptr : src0_ptr;
            const int ld = ld_second ? src1_ld : src0_ld;
            const int tstride = ld_second ? src1_ts : src0_ts;
            ld_tap_off = u_dt * tstride + (u_dy * c_Win + u_dx) * ld + (u_cc - (ld_second ? src0_units : 0)) * 8 + (f4 & 1) * 4;
        }
    };
    auto gather = [&](int j) {
        if constexpr (ALIGNED) {
            const bool ok = (vmask[j] & s_sel) == s_sel;
            int voff = s_second ? voff1[j] : voff0[j];
            voff = ok ? voff : static_cast<int>(0x80000000u);               // beyond the descriptor: reads as zero
            return to_float4(__builtin_amdgcn_raw_buffer_load_b128(s_rsrc, voff, s_off, 0));
        } else {
            const int iy = py0[j] + u_dy, ix = px0[j] + u_dx;
            // (bitwise &: a short-circuit && would come back as a branch around the rest)
            const bool ok = ld_valid & pvalid[j] & (static_cast<unsigned>(iy) < static_cast<unsigned>(c_Hin)) &
                            (static_cast<unsigned>(ix) < static_cast<unsigned>(c_Win)) & ((ptmin[j] + u_dt) >= 0);
            const float* src = ok ? ld_base + ((ld_second ? poff1[j] : poff0[j]) + ld_tap_off) : zero_page;
            return *reinterpret_cast<const float4*>(src);
        }
    };
    auto load_a = [&](int j) {
        if (j == 0) areg0 = gather(0);
        else if (j == 1) areg1 = gather(1);
        else if (j == 2) areg2 = gather(NA > 2 ? 2 : 0);
        else areg3 = gather(NA > 2 ? 3 : 0);
    };
    auto load_b = [&](int k) {
        float4 v;
        if constexpr (ALIGNED) {
            v = to_float4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, k == BLOADS - 1 ? w_voff_last : w_voff, w_soff + k * 4096, 0));
        } else {
            const float* wsrc = ld_stage < c_k_chunks ? wnext : wfirst;  // past the end: any valid address
            v = *reinterpret_cast<const float4*>(wsrc + k * 1024);
        }
        if (k == 0) breg0 = v;
        else if (k == 1) breg1 = v;
        else if (k == 2) breg2 = v;
        else breg3 = v;
    };
    auto advance = [&]() {
        ++ld_stage;
        if constexpr (ALIGNED) {                          // all on the scalar unit
            ++s_g;
            const bool cg = s_g == s_groups;
            s_g = cg ? 0 : s_g;
            s_tap += cg ? 1 : 0;
            s_dx += cg ? 1 : 0;
            const bool cx = s_dx == c_kW;
            s_dx = cx ? 0 : s_dx;
            s_dy += cx ? 1 : 0;
            const bool cy = s_dy == c_kH;
            s_dy = cy ? 0 : s_dy;
            s_dt += cy ? 1 : 0;
        } else {
            wnext += BK * BN;
            u_cc += 4;                                    // one stage = 4 units, carrying into the tap and its (dt, dy, dx)
            if constexpr (SMALLCIN) {
                while (u_cc >= c_cin_units) {             // fewer than four units per tap: several carries per stage
                    u_cc -= c_cin_units;
                    ++u_tap;
                    if (++u_dx == c_kW) {
                        u_dx = 0;
                        if (++u_dy == c_kH) {
                            u_dy = 0;
                            ++u_dt;
                        }
                    }
                }
            } else {                                      // cin_units >= 4: at most one carry, as selects
                const bool carry = u_cc >= c_cin_units;
                u_cc -= carry ? c_cin_units : 0;
                u_tap += carry ? 1 : 0;
                u_dx += carry ? 1 : 0;
                const bool cx = u_dx == c_kW;
                u_dx = cx ? 0 : u_dx;
                u_dy += cx ? 1 : 0;
                const bool cy = u_dy == c_kH;
                u_dy = cy ? 0 : u_dy;
                u_dt += cy ? 1 : 0;
            }
        }
    };
    // LDS addresses: thread constants (floats from smem) + compile-time offsets
    const int a_st = prow * BK + ((f4 ^ ((prow >> 1) & 7)) << 2);                     // + 32 j BK + buf BM BK
    const int b_st = W_BASE + tid * 4;                                                 // + 1024 k + buf W_STAGE
    const int a_row = wm * (32 * MT) + m;
    int a_rd[4];                                                                       // + 32 t BK + buf BM BK
#pragma unroll
    for (int q = 0; q < 4; ++q) a_rd[q] = a_row * BK + (((2 * q + hi) ^ ((a_row >> 1) & 7)) << 2);
    const int b_rd = W_BASE + (hi * BN + wn * (32 * NT) + m) * 4;                      // + (2 q BN + 32 nt) 4 + buf W_STAGE
    // bf16: a lane's eight k of MFMA kh (k = 16 kh + 8 hi ..) are slot 2 kh + hi of its row of the bf16 image
    int a_rd16[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) a_rd16[kh] = a_row * 16 + (((2 * kh + hi) ^ ((a_row >> 2) & 3)) << 2);     // + 32 t 16 + buf A_STAGE
    // bf16 image of the A tile: 16 floats' worth of bytes per row; this thread's four k are half of slot f4 >> 1
    const int a_st16 = prow * 16 + (((f4 >> 1) ^ ((prow >> 2) & 3)) << 2) + (f4 & 1) * 2;     // + 32 j 16 + buf A_STAGE
    auto store_a = [&](int buf, int j) {
        const float4 v = j == 0 ? areg0 : j == 1 ? areg1 : j == 2 ? areg2 : areg3;
        if constexpr (SPLIT_READ) *reinterpret_cast<float4*>(&smem[a_st + 32 * j * BK + buf * A_STAGE]) = v;
        else if constexpr (SPLIT) {
            // x = t1 + t2 + t3: the remainders are exact in fp32
            auto minus = [](const float4& x, const float4& y) { return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); };
            const uint2 t1 = pack_bf16x4(v);
            const float4 r1 = minus(v, unpack_bf16x4(t1));
            const uint2 t2 = pack_bf16x4(r1);
            const float4 r2 = minus(r1, unpack_bf16x4(t2));
            uint2* const at = reinterpret_cast<uint2*>(&smem[a_st16 + 32 * j * 16 + buf * A_STAGE]);
            at[0] = t1;
            at[A_PLANE / 2] = t2;
            at[A_PLANE] = pack_bf16x4(r2);
        } else if constexpr (BF16) *reinterpret_cast<uint2*>(&smem[a_st16 + 32 * j * 16 + buf * A_STAGE]) = pack_bf16x4(v);
        else *reinterpret_cast<float4*>(&smem[a_st + 32 * j * BK + buf * A_STAGE]) = v;
    };
    auto store_b = [&](int buf, int k) {
        // (bf16, BN = 32: a stage's weights are 2 KiB - the upper half of the threads fetched nothing and has no slot)
        if (BF16 && BN == 32 && tid >= 128 && (!SPLIT || k == BLOADS - 1)) return;
        *reinterpret_cast<float4*>(&smem[b_st + 1024 * k + buf * W_STAGE]) = k == 0 ? breg0 : k == 1 ? breg1 : k == 2 ? breg2 : breg3;
    };
    auto lds_a = [&](int buf, int q, int t) {
        return *reinterpret_cast<const float4*>(&smem[a_rd[q] + 32 * t * BK + buf * A_STAGE]);
    };
    auto lds_b = [&](int buf, int q, int nt) {
        return *reinterpret_cast<const float4*>(&smem[b_rd + (2 * q * BN + 32 * nt) * 4 + buf * W_STAGE]);
    };
    // piece i of a stage's side work, i = 0 .. N_PIECES-1; the stage running out of `buf` fills the other buffer
    auto side_piece = [&](int buf, int i) {
#if FIERY_CONV_EARLY_LOADS
        // every register is requested again right after the store that frees it: a request then has the whole stage to land
        // (with all stores first and all requests after them it had the second half of one stage and the first MFMAs of the next)
        if (i < 3) load_setup(i);
        else if (i < 3 + 2 * NA) {
            if ((i - 3) & 1) load_a((i - 3) >> 1);
            else store_a(buf ^ 1, (i - 3) >> 1);
        } else if (i < 3 + 2 * NA + 2 * BLOADS) {
            if ((i - 3 - 2 * NA) & 1) load_b((i - 3 - 2 * NA) >> 1);
            else store_b(buf ^ 1, (i - 3 - 2 * NA) >> 1);
        } else advance();
        return;
#endif
        constexpr int S0 = NA + BLOADS;            // first setup piece
        if (i < NA) store_a(buf ^ 1, i);
        else if (i < S0) store_b(buf ^ 1, i - NA);
        else if (i < S0 + 3) load_setup(i - S0);
        else if (i < S0 + 3 + NA) load_a(i - (S0 + 3));
        else if (i < S0 + 3 + NA + BLOADS) load_b(i - (S0 + 3 + NA));
        else advance();
    };


    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if constexpr (CLK) {
        if (clk_trace) clk_trace[1] = wall_clock64();
        if (tid == 0 && (bid_x & 15) == 0) {
            clk_c0 = clock64();
            clk_w0 = wall_clock64();
        }
    }
    v16f acc[MT * NT];
#pragma unroll
    for (int t = 0; t < MT * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if constexpr (HALO) {
        // ---- halo loop: 3 x 3, stride 1, 'same' padding, bf16 operands ------------------------------------------------
        // The scalar-addressed loop gathers the A tile once per (tap, 32-channel group): nine times the same pixels, shifted.
        // With bf16's short MFMAs that traffic - 1 KB from L2 per MFMA - is what bounds the loop.  Here the loop runs
        // channel group by channel group: the tile's 64 consecutive pixels and their neighbours are fetched ONCE per group
        // as three runs of 66 pixels (the rows above, of, and below the tile's pixels: linear pixel index - W - 1 .., - 1 ..,
        // + W - 1 ..), rounded to bf16 and laid out in LDS, and the nine taps are nine stages that read shifted windows of
        // that block: tap (dy, dx) of tile pixel i is entry i + dx of run dy.  Only the weights change from stage to stage.
        //   * rows outside the image: an entry of run 0 (2) whose pixel lies in the last (first) row of an image belongs to
        //     no pixel of THIS image's window - it is fetched as zero (offset past the descriptor), like pixels outside the
        //     tensor; that settles dy.
        //   * columns: the left neighbour of a pixel with x = 0 is the previous row's last pixel in linear order; such a
        //     lane reads the run's zero entry instead (one address select per lane, made once: rdL / rdR below).
        //   * entries are 80 bytes apart (64 of data): 16 consecutive entries at one 16-byte slot then fall into 16
        //     different slots of the 256-byte bank row for every lane group of ds_read_b128, and every tap's address is a
        //     per-lane base plus an immediate.
        constexpr int RUN = BM + HALO_RUN_EXTRA;                    // entries per run: BM + 2 pixels, then zeros
        constexpr int PITCH = halo_pitch(BF16);                     // floats between entries
        constexpr int NQ = BF16 ? 2 : 4;                            // 16-byte operand pieces per lane, entry / cout and stage
        constexpr int W_CHUNK = BK * BN * (BF16 ? 2 : 4);           // bytes of one stage of the packed weights
        constexpr int NE = (3 * (BM + 2) + 31) / 32;                // halo elements (16 bytes of fp32) per thread and group
        static_assert(3 * RUN * PITCH <= SMEM_FLOATS && NT == 1 && (MT == 1 || MT == 2 || MT == 4), "halo block must fit; one cout tile per wavefront");
        const int W_img = p.Wout, H_img = p.Hout;
        const int groups = p.cin_units >> 2, groups0 = p.src[0].units >> 2;
        const int f4h = tid & 7, prow_h = tid >> 3;
        int hv0[NE], hv1[NE], hst[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int ent = prow_h + 32 * i;
            const int r = ent >= 2 * (BM + 2) ? 2 : ent >= (BM + 2) ? 1 : 0;
            const int idx = ent - r * (BM + 2);
            const int q = pix0 + (r - 1) * W_img - 1 + idx;
            const bool inside = ent < 3 * (BM + 2) && q >= 0 && q < M;
            const int qq = inside ? q : 0;
            const int o = fast_div(qq, p.mg_hw, p.sh_hw);
            const int ppi = qq - o * HWout;
            const int y = fast_div(ppi, p.mg_w, p.sh_w);
            const int b = fast_div(o, p.mg_t, p.sh_t), tl = o - b * p.Tout;
            const bool ok = inside && (r == 0 ? y <= H_img - 2 : r == 2 ? y >= 1 : true);
            const int kofs = (f4h >> 1) * 8 + (f4h & 1) * 4;
            const int e0 = 4 * (b * static_cast<int>(p.src[0].bstride) + (tl + p.tinadd) * static_cast<int>(p.src[0].tstride) + ppi * p.src[0].ld + kofs);
            const int e1 = 4 * (b * static_cast<int>(p.src[1].bstride) + (tl + p.tinadd) * static_cast<int>(p.src[1].tstride) + ppi * p.src[1].ld + kofs);
            hv0[i] = ok ? e0 : static_cast<int>(0x80000000u);
            hv1[i] = ok ? e1 : static_cast<int>(0x80000000u);
            // where the element goes (ent + r = r RUN + idx): bf16 - 8-byte units, half of slot f4 >> 1; fp32 - 16-byte units, slot f4
            hst[i] = BF16 ? (ent + r) * (PITCH / 2) + (f4h >> 1) * 2 + (f4h & 1) : (ent + r) * (PITCH / 4) + f4h;
        }
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src[0].ptr), 0, p.src[0].ext_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src[1].ptr ? p.src[1].ptr : p.src[0].ptr), 0,
                                                                               p.src[1].ptr ? p.src[1].ext_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.w) + static_cast<long long>(tile_n) * p.k_chunks * W_CHUNK), 0,
            p.k_chunks * W_CHUNK, 0x00020000);
        float4 hreg[NE];
        auto halo_load = [&](int g) {
            const bool second = g >= groups0;
            const int soff = 128 * (g - (second ? groups0 : 0));
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                float4 v;
                if (second) v = to_float4(__builtin_amdgcn_raw_buffer_load_b128(rsrc1, hv1[i], soff, 0));
                else v = to_float4(__builtin_amdgcn_raw_buffer_load_b128(rsrc0, hv0[i], soff, 0));
                hreg[i] = v;
            }
        };
        const bool last_live = prow_h + 32 * (NE - 1) < 3 * (BM + 2);     // (all but a thread's last element always exist)
        uint2* const smem2 = reinterpret_cast<uint2*>(smem);
        float4* const smem4 = reinterpret_cast<float4*>(smem);
        auto halo_store = [&]() {
#pragma unroll
            for (int i = 0; i < NE; ++i)
                if (i < NE - 1 || last_live) {
                    if constexpr (BF16) {
                        smem2[hst[i]] = pack_bf16x4(hreg[i]);
                    } else {                                         // (recomputed: seven registers fewer held across the loop)
                        const int ent = prow_h + 32 * i;
                        const int r = ent >= 2 * (BM + 2) ? 2 : ent >= (BM + 2) ? 1 : 0;
                        smem4[(ent + r) * (PITCH / 4) + f4h] = hreg[i];
                    }
                }
        };
        // The weights never touch LDS: a lane's B operand of an MFMA - eight k of one cout - is one 16-byte piece of the
        // packed image [k / 8][cout][k % 8], and each wavefront multiplies its own 32 couts, so it requests exactly its
        // operands (two 16-byte loads per lane and stage) straight into registers, two stages ahead, in a ring of three
        // register sets (nine stages per group = three turns of the ring: the set of a stage is a compile-time index).
        // No LDS writes, no W reads, and no barrier inside a channel group: the wavefronts only meet when the halo changes.
        const int w_vo = (hi * BN + wn * 32 + m) * 16;
        // (fp32: four pieces per stage would make the ring of three stages 48 registers; its pieces go through a ring of six
        // instead, each requested five pieces - a stage and a quarter - before it is multiplied: 36 pieces per group = six turns)
        constexpr int NP = 9 * NQ, RING = 6, AHEAD = 5;
        float4 wp[RING];
        auto w_piece = [&](int slot, int g, int pidx) {
            const int gg = pidx >= NP ? g + 1 : g, pi = pidx >= NP ? pidx - NP : pidx;
            int chunk = (pi / NQ) * groups + gg;
            chunk = chunk < p.k_chunks ? chunk : p.k_chunks - 1;
            wp[slot] = to_float4(__builtin_amdgcn_raw_buffer_load_b128(wrs, w_vo, chunk * W_CHUNK + (pi % NQ) * (2 * BN * 16), 0));
        };
        float4 wq[3][NQ];
        auto w_load = [&](int set, int g, int t) {              // stage (t, g) of the packed weights: chunk t groups + g
            int chunk = t * groups + g;
            chunk = chunk < p.k_chunks ? chunk : p.k_chunks - 1;                        // past the end: any chunk
            const int soff = chunk * W_CHUNK;
#pragma unroll
            for (int q = 0; q < NQ; ++q) wq[set][q] = to_float4(__builtin_amdgcn_raw_buffer_load_b128(wrs, w_vo, soff + q * (2 * BN * 16), 0));
        };
        // this lane's windows: tile pixels pl (+ 32 for the second pixel tile), slot hi of each entry; the zero entry for the
        // taps that would wrap around a row end
        int rdC[MT], rdL[MT], rdR[MT];
#pragma unroll
        for (int tt = 0; tt < MT; ++tt) {
            const int pl_h = wm * (32 * MT) + tt * 32 + m;
            const int gp = pix0 + pl_h;
            const int g_ = gp < M ? gp : 0;
            const int o = fast_div(g_, p.mg_hw, p.sh_hw);
            const int ppi = g_ - o * HWout;
            const int y = fast_div(ppi, p.mg_w, p.sh_w), x = ppi - y * W_img;
            const int rdZ = (BM + 2) * PITCH + hi * 4;
            rdC[tt] = pl_h * PITCH + hi * 4;
            rdL[tt] = x == 0 ? rdZ : rdC[tt];
            rdR[tt] = x == W_img - 1 ? rdZ - 2 * PITCH : rdC[tt];
        }
        // prologue: zero entries, group 0's halo; group 1's halo and the first two stages' weights stay in flight
        if (tid < 3 * PITCH) smem[((tid / PITCH) * RUN + BM + 2) * PITCH + tid % PITCH] = 0.f;
        halo_load(0);
        halo_store();
        if constexpr (BF16) {
            w_load(0, 0, 0);
            w_load(1, 0, 1);
        } else {
#pragma unroll
            for (int pi = 0; pi < AHEAD; ++pi) w_piece(pi, 0, pi);
        }
        if (groups > 1) halo_load(1);
        __syncthreads();
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                int rd[MT];
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) rd[tt] = (dx == 0 ? rdL[tt] : dx == 2 ? rdR[tt] : rdC[tt]) + (dy * RUN + dx) * PITCH;
                if constexpr (BF16) {
                    bf16x8 a8[MT][2];
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) {
                        a8[tt][0] = load_bf16x8(&smem[rd[tt]]);
                        a8[tt][1] = load_bf16x8(&smem[rd[tt] + 8]);
                    }
                    const bf16x8 b0 = bits_bf16x8(wq[t % 3][0]), b1 = bits_bf16x8(wq[t % 3][1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) {
                        acc[tt] = mfma_bf16_32x32x16(a8[tt][0], b0, acc[tt]);
                        acc[tt] = mfma_bf16_32x32x16(a8[tt][1], b1, acc[tt]);
                    }
                } else {
                    // fp32: a lane's piece q holds its row's (its cout's) k = 8 q + 4 hi + j, j < 4: four MFMA k-steps; piece
                    // q + 1 is read from LDS while the MFMAs of piece q run
                    float4 a_cur[MT], a_nxt[MT];
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) a_cur[tt] = *reinterpret_cast<const float4*>(&smem[rd[tt]]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q < 3) {
#pragma unroll
                            for (int tt = 0; tt < MT; ++tt) a_nxt[tt] = *reinterpret_cast<const float4*>(&smem[rd[tt] + (q + 1) * 8]);
                        }
                        const float4 b4 = wp[(t * NQ + q) % RING];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bv = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
#pragma unroll
                            for (int tt = 0; tt < MT; ++tt) {
                                const float av = j == 0 ? a_cur[tt].x : j == 1 ? a_cur[tt].y : j == 2 ? a_cur[tt].z : a_cur[tt].w;
                                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tt], 0, 0, 0);
                            }
                        }
                        if (q < 3) {
#pragma unroll
                            for (int tt = 0; tt < MT; ++tt) a_cur[tt] = a_nxt[tt];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        w_piece((t * NQ + q + AHEAD) % RING, g, t * NQ + q + AHEAD);      // the slot the previous piece just left
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF16) {
                    // the set this stage leaves takes the stage after next: (t + 2, g), wrapping into the next group
                    if (t + 2 < 9) w_load((t + 2) % 3, g, t + 2);
                    else w_load((t + 2) % 3, g + 1, t + 2 - 9);
                }
            }
            if (g + 1 < groups) {
                __syncthreads();                                   // every wavefront has read its last window of this group
                halo_store();                                      // (the next group's halo was requested a group ago)
                __syncthreads();
                if (g + 2 < groups) halo_load(g + 2);
            }
        }
        __syncthreads();                                           // the epilogue stages its tile over the halo
    } else {
    // ---- software pipeline ------------------------------------------------------------------------------------
    // While stage s is multiplied out of LDS buffer s&1, stage s+1 sits in registers (requested one iteration
    // earlier, so its latency is long gone) and is written to the other buffer, and stage s+2 is requested into
    // the registers that frees.  The pieces are dealt out evenly between the MFMAs of the running stage.
    // prologue: stage 0 -> LDS buffer 0, stage 1 -> registers (it stays in flight across the barrier)
#pragma unroll
    for (int k = 0; k < 3; ++k) load_setup(k);
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j);
#pragma unroll
    for (int k = 0; k < BLOADS; ++k) load_b(k);
    advance();
#pragma unroll
    for (int i = 0; i < N_PIECES; ++i) side_piece(1, i);
    __syncthreads();

    constexpr int N_MFMA = (BF16 ? (SPLIT ? 12 : 2) : 16) * MT * NT;
    auto stage_body = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        if constexpr (SPLIT) {
            // per sixteen k and 32 x 32 block the six partial products, smallest first: a1 w3, a3 w1, a2 w2, a1 w2, a2 w1, a1 w1
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TW[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 a8[MT][3], b8[NT][3];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    if constexpr (SPLIT_READ) {
                        // the lane's eight k (16 kh + 8 hi ..) are slots 4 kh + 2 hi, + 1 of its row of the fp32 image
                        const float4 lo = *reinterpret_cast<const float4*>(&smem[a_row * BK + (((4 * kh + 2 * hi) ^ ((a_row >> 1) & 7)) << 2) + 32 * t * BK + buf * A_STAGE]);
                        const float4 up = *reinterpret_cast<const float4*>(&smem[a_row * BK + (((4 * kh + 2 * hi + 1) ^ ((a_row >> 1) & 7)) << 2) + 32 * t * BK + buf * A_STAGE]);
                        const float x[8] = {lo.x, lo.y, lo.z, lo.w, up.x, up.y, up.z, up.w};
                        split_bf16x8(x, a8[t][0], a8[t][1], a8[t][2]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e) a8[t][e] = load_bf16x8(&smem[a_rd16[kh] + 32 * t * 16 + e * A_PLANE + buf * A_STAGE]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        b8[nt][e] = load_bf16x8(&smem[W_BASE + buf * W_STAGE + e * W_PLANE + ((2 * kh + hi) * BN + wn * (32 * NT) + nt * 32 + m) * 4]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            if constexpr (PIXLANE) acc[t * NT + nt] = mfma_bf16_32x32x16(b8[nt][TW[k6]], a8[t][TA[k6]], acc[t * NT + nt]);
                            else acc[t * NT + nt] = mfma_bf16_32x32x16(a8[t][TA[k6]], b8[nt][TW[k6]], acc[t * NT + nt]);
                            const int s = ((kh * 6 + k6) * NT + nt) * MT + t;
#pragma unroll
                            for (int i = 0; i < N_PIECES; ++i)
                                if ((i * N_MFMA) / N_PIECES == s) side_piece(buf, i);
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
            __syncthreads();
            return;
        }
        if constexpr (BF16) {
            // two MFMAs of sixteen k per 32 x 32 block and stage; the W image is [k / 8][cout][k % 8] bf16 and the A image
            // [pixel][k] bf16, so each operand of a lane is one 16-byte read
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 a8[MT], b8[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a8[t] = load_bf16x8(&smem[a_rd16[kh] + 32 * t * 16 + buf * A_STAGE]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b8[nt] = load_bf16x8(&smem[W_BASE + buf * W_STAGE + ((2 * kh + hi) * BN + wn * (32 * NT) + nt * 32 + m) * 4]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        if constexpr (PIXLANE) acc[t * NT + nt] = mfma_bf16_32x32x16(b8[nt], a8[t], acc[t * NT + nt]);
                        else acc[t * NT + nt] = mfma_bf16_32x32x16(a8[t], b8[nt], acc[t * NT + nt]);
                        const int s = (kh * NT + nt) * MT + t;
#pragma unroll
                        for (int i = 0; i < N_PIECES; ++i)
                            if ((i * N_MFMA) / N_PIECES == s) side_piece(buf, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            __syncthreads();
            return;
        }
        // operands of k-group q+1 are read from LDS while the MFMAs of group q run (one b128 per 32x4 operand
        // block: A rows are [pixel][k], the W image is [k/4][cout][k%4])
        float4 a_cur[MT], b_cur[NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) a_cur[t] = lds_a(buf, 0, t);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt] = lds_b(buf, 0, nt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a_nxt[MT], b_nxt[NT];
            if (q < 3) {
#pragma unroll
                for (int t = 0; t < MT; ++t) a_nxt[t] = lds_a(buf, q + 1, t);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = lds_b(buf, q + 1, nt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // this lane's k for the step: 8q + 4hi + j, for its A element and its W element alike
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = j == 0 ? b_cur[nt].x : j == 1 ? b_cur[nt].y : j == 2 ? b_cur[nt].z : b_cur[nt].w;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const float av = j == 0 ? a_cur[t].x : j == 1 ? a_cur[t].y : j == 2 ? a_cur[t].z : a_cur[t].w;
                        if constexpr (PIXLANE) acc[t * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[t * NT + nt], 0, 0, 0);
                        else acc[t * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t * NT + nt], 0, 0, 0);
                        // MFMA slot s of the stage is followed by the side-work pieces dealt to it (spread evenly)
                        const int s = ((q * 4 + j) * NT + nt) * MT + t;
#pragma unroll
                        for (int i = 0; i < N_PIECES; ++i)
                            if ((i * N_MFMA) / N_PIECES == s) side_piece(buf, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (q < 3) {
#pragma unroll
                for (int t = 0; t < MT; ++t) a_cur[t] = a_nxt[t];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_cur[nt] = b_nxt[nt];
            }
        }
        __syncthreads();
    };
    {
        const int n_chunks = SK ? sk_ke - sk_kb : c_k_chunks;
        int chunk = 0;
        for (; chunk + 1 < n_chunks; chunk += 2) {
            stage_body(std::integral_constant<int, 0>{});
            stage_body(std::integral_constant<int, 1>{});
        }
        if (chunk < n_chunks) stage_body(std::integral_constant<int, 0>{});
    }
    }      // !HALO
    if constexpr (SK) {
        // ---- stream-K: a tile whose chunks are shared between workgroups ----------------------------------------------------
        // Every holder of a part writes its accumulators to its workspace slot - in accumulator order, 16 bytes per lane and
        // piece, so the stores are unit-stride and nothing is staged - and takes a ticket on the tile's counter; the LAST arrival
        // reads all parts back IN PART ORDER (its own too: the sum must not depend on who came last), and runs the epilogue on
        // the sum.  The parts may sit on different XCDs, whose L2s are not coherent with each other: partial stores and loads
        // carry sc0 sc1 (system scope: written through / fetched past the L2s; tools/probe/xcd_partials_probe.hip measured
        // them at the plain accesses' speed, where agent-scope fences cost 4x - profiles/r5_xcd_partials_probe.txt), and the
        // ticket is a memory-side atomic.
        if (!(sk_kb == 0 && sk_ke == c_k_chunks)) {
            const long long total = static_cast<long long>(p.sk_tiles) * c_k_chunks;
            const int nwg = sk_nwg;
            auto owner = [&](long long u) { return static_cast<int>(((u + 1) * nwg - 1) / total); };       // the workgroup holding unit u
            const long long t0 = static_cast<long long>(sk_tile) * c_k_chunks;
            const int j0 = owner(t0), j1 = owner(t0 + c_k_chunks - 1);
            // a workgroup holds at most two partial tiles: its first (slot 0) and its last (slot 1)
            auto slot_of = [&](int j) {
                const long long b = static_cast<long long>(j) * total / nwg;
                return 2 * j + (static_cast<int>(b / c_k_chunks) == sk_tile ? 0 : 1);
            };
            auto part_rsrc = [&](int j) {
                return __builtin_amdgcn_make_buffer_rsrc(p.sk_ws + static_cast<long long>(slot_of(j)) * (BM * BN), 0, BM * BN * 4, 0x00020000);
            };
            {
                const __amdgpu_buffer_rsrc_t mine = part_rsrc(sk_j);
#pragma unroll
                for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                        decltype(__builtin_amdgcn_raw_buffer_load_b128(mine, 0, 0, 0)) raw;
                        __builtin_memcpy(&raw, &v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, mine, ((i * 4 + g) * 256 + tid) * 16, 0, 17);      // sc0 | sc1
                    }
            }
            vmem_done();                                              // this wavefront's stores are out before the barrier lets the ticket go
            __syncthreads();
            int* const flag = reinterpret_cast<int*>(smem);           // (the stages are free: the loop's last barrier has passed)
            if (tid == 0) *flag = atomicAdd(p.sk_cnt + sk_tile, 1);
            __syncthreads();
            const int ticket = *flag;
            __syncthreads();                                          // (read by everyone before the epilogue stages its tile there)
            if (ticket != j1 - j0) return;                            // not the last arrival: this part is done
#pragma unroll
            for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            for (int j = j0; j <= j1; ++j) {
                const __amdgpu_buffer_rsrc_t part = part_rsrc(j);
#pragma unroll
                for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = to_float4(__builtin_amdgcn_raw_buffer_load_b128(part, ((i * 4 + g) * 256 + tid) * 16, 0, 17));
                        acc[i][4 * g] += v.x;  acc[i][4 * g + 1] += v.y;  acc[i][4 * g + 2] += v.z;  acc[i][4 * g + 3] += v.w;
                    }
            }
            if (tid == 0) atomicExch(p.sk_cnt + sk_tile, 0);          // ready for the next launch
        }
    }
    // The epilogue's vector and memory instructions share the SIMD with the other workgroups' MFMAs and, at equal priority,
    // are served in the gaps those leave: an epilogue of 5 us alone takes 27 us there (profiles/r4_conv_phases.txt), a time
    // during which this wavefront feeds the matrix pipe nothing.  Raised priority gets it through and back to MFMAs.
#if FIERY_CONV_EPILOGUE_PRIO
    __builtin_amdgcn_s_setprio(FIERY_CONV_EPILOGUE_PRIO);
#endif
    if constexpr (CLK) {
        // effective shader clock under this kernel's load = cycles / ticks * 100 MHz
        if (tid == 0 && (bid_x & 15) == 0 && g_clk_probe) {
            atomicAdd(g_clk_probe, static_cast<unsigned long long>(clock64() - clk_c0));
            atomicAdd(g_clk_probe + 1, static_cast<unsigned long long>(wall_clock64() - clk_w0));
            atomicAdd(g_clk_probe + 2, static_cast<unsigned long long>(clk_c0 - clk_entry));       // prologue (set-up + first loads)
            atomicAdd(g_clk_probe + 6, 1ull);
        }
    }
    const unsigned long long clk_loop_end = CLK ? clock64() : 0ull;
    if constexpr (CLK) {
        if (clk_trace) clk_trace[2] = wall_clock64();
    }
    auto clk_finish = [&]() {                                  // epilogue cycles of the sampled workgroups (staged paths)
        if constexpr (CLK) {
            if (clk_trace) clk_trace[3] = wall_clock64();
            if (tid == 0 && (bid_x & 15) == 0 && g_clk_probe)
                atomicAdd(g_clk_probe + 3, static_cast<unsigned long long>(clock64() - clk_loop_end));
        }
    };

    // ---- staged epilogue (plain mode): accumulators -> LDS tile [pixel][cout] -> 16-byte rows -----------------
    // A lane holds one cout for 16 scattered pixel rows, so storing from registers means 4-byte accesses 128 B
    // at a time; going through LDS turns the tile into full rows: every residual load and output store is a
    // 16-byte, unit-stride access.  `store_rows(width, ...)` finishes a staged tile of `width` couts.
    int res_pre_rows = p.res_pre;                              // (the argument block itself is read-only)
    auto store_rows = [&](int width, int cout0, const float* scale, const float* shift, int act, bool with_bias,
                          const float4* fetched = nullptr) {
        const int c4n = width >> 2;                                 // 16-byte chunks per pixel row
        const int rows_per_pass = 256 / c4n;
        const int c4 = tid % c4n, prow0 = tid / c4n;
        const int co = cout0 + c4 * 4;
        const int half = p.cout_pad >> 1;
        const bool gates_upper = p.epi == FIERY_EPI_GRU_GATES && co >= half;
        // padding couts are never stored (gate epilogue: cout_store channels of EACH half - the two gates)
        if ((gates_upper ? co - half : co) >= p.cout_store) return;
        // (`fetched`: the caller requested this thread's scale / shift before it staged the tile - one memory round trip less
        // between the K loop's end and the first row)
        const float4 sc = fetched ? fetched[0] : *reinterpret_cast<const float4*>(scale + co);
        const float4 sh = fetched ? fetched[1] : *reinterpret_cast<const float4*>(shift + co);
        if constexpr (CLK) {                                              // scale / shift have arrived
            if (clk_trace) clk_trace[7] = wall_clock64() + (__builtin_bit_cast(int, sc.x) & __builtin_bit_cast(int, sh.x) & 0);
        }
        if (p.vec_epilogue & 2) {
            // DENSE TENSORS (round 4; the host sets the bit when every tensor the epilogue touches has its images back to back -
            // img_stride == Hout Wout ld - and spans less than 2 GB): pixel gp of a tensor then sits gp * ld floats from its base
            // whatever image it belongs to, so a row costs one 32-bit add per tensor on a buffer descriptor (whose range check also
            // drops the rows past M) instead of an image / in-image split with its wrap-around loop and three 64-bit
            // multiply-adds.  The epilogue's ~1,000 instructions per wavefront are served in the gaps other wavefronts' MFMAs
            // leave: every instruction less is ~60 cycles sooner back in the K loop.
            const int m_rows = M;
            auto rsrc = [&](const float* ptr, int ld_) {
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, ptr ? m_rows * ld_ * 4 : 0, 0x00020000);
            };
            auto load4 = [&](__amdgpu_buffer_rsrc_t r, int voff) {
                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
                float4 f;
                __builtin_memcpy(&f, &raw, 16);
                return f;
            };
            auto store4 = [&](__amdgpu_buffer_rsrc_t r, int voff, const float4& f) {
                decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) raw;
                __builtin_memcpy(&raw, &f, 16);
                __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, 0, 0);
            };
            // The row loop is instantiated per kind of epilogue (round 4): with the kind read at run time every row paid for
            // all of them - two packed adds of a bias that is mostly absent, the residual added before AND after the
            // activation with four selects each, a NaN-quieting max in front of the ReLU, scalar branches on the activation -
            // ~30 vector instructions where BatchNorm + ReLU needs six; and a vector instruction of an epilogue costs ~60
            // cycles next to other workgroups' MFMAs (above).  -1 = read at run time (the general loop, everything else).
            auto rows = [&](auto epi_c, auto act_c, auto res_c, auto bias_c) {
                constexpr int kEpi = decltype(epi_c)::value, kAct = decltype(act_c)::value;
                constexpr int kRes = decltype(res_c)::value;               // 0 none, 1 before the activation, 2 after it
                constexpr int kBias = decltype(bias_c)::value;             // 0 none, 1 per-image bias rows
                const int epi_ = kEpi < 0 ? p.epi : kEpi;
                const int act_ = kAct < 0 ? act : kAct;
                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                int gp = pix0 + prow0;
                const bool plain = epi_ == FIERY_EPI_PLAIN, gates = epi_ == FIERY_EPI_GRU_GATES;
                // descriptors are wave-uniform (one per tensor); which of them a lane uses - a gate launch's lower channel half
                // stores the update gate to `out`, its upper half reads the state and stores to `out2` - is a per-lane predicate
                const bool upper = gates && gates_upper;
                const TensP& t_a = plain ? p.res : p.aux0;
                const int c_x = upper ? co - half : co;                          // channel inside out2 / aux0 for the upper half
                const bool has_a = plain ? (kRes < 0 ? p.res.ptr != nullptr : kRes != 0) : (gates ? upper : true);
                const bool res_first = kRes < 0 ? res_pre_rows != 0 : kRes == 1;
                const bool has_b = !plain && !gates;                              // GRU_OUT: aux1 = state
                const bool has_d1 = has_b && p.out2.ptr != nullptr;
                const __amdgpu_buffer_rsrc_t r_out = rsrc(p.out.ptr, p.out.ld), r_out2 = rsrc(p.out2.ptr, p.out2.ld),
                                             r_a = rsrc(t_a.ptr, t_a.ld), r_b = rsrc(has_b ? p.aux1.ptr : nullptr, p.aux1.ld);
                int o_out = (gp * p.out.ld + co) * 4, o_out2 = (gp * p.out2.ld + c_x) * 4, o_a = (gp * t_a.ld + c_x) * 4,
                    o_b = (gp * p.aux1.ld + co) * 4;
                const int s_out = rows_per_pass * p.out.ld * 4, s_out2 = rows_per_pass * p.out2.ld * 4, s_a = rows_per_pass * t_a.ld * 4,
                          s_b = rows_per_pass * p.aux1.ld * 4;
                const bool bias_rows = kBias < 0 ? (with_bias && p.img_bias) : kBias != 0;
                auto bias_of = [&](int g) {
                    return *reinterpret_cast<const float4*>(p.img_bias + fast_div(g, p.mg_hw, p.sh_hw) * p.cout_pad + co);
                };
                float4 cur_a = has_a ? load4(r_a, o_a) : zero4, cur_b = has_b ? load4(r_b, o_b) : zero4;
                float4 cur_bias = (bias_rows && gp < M) ? bias_of(gp) : zero4;
                for (int pl = prow0; pl < BM; pl += rows_per_pass) {
                    if (gp >= M) break;
                    const int gp_n = gp + rows_per_pass;
                    const bool more = pl + rows_per_pass < BM && gp_n < M;
                    // the next row's operands are requested before this row is stored (an in-place residual is a different pixel)
                    const float4 nxt_a = (has_a && more) ? load4(r_a, o_a + s_a) : zero4;
                    const float4 nxt_b = (has_b && more) ? load4(r_b, o_b + s_b) : zero4;
                    const float4 nxt_bias = (bias_rows && more) ? bias_of(gp_n) : zero4;
                    float4 v = *reinterpret_cast<const float4*>(&smem[pl * width + c4 * 4]);
                    if (kBias != 0) { v.x += cur_bias.x;  v.y += cur_bias.y;  v.z += cur_bias.z;  v.w += cur_bias.w; }
                    v.x = fmaf(v.x, sc.x, sh.x);  v.y = fmaf(v.y, sc.y, sh.y);  v.z = fmaf(v.z, sc.z, sh.z);  v.w = fmaf(v.w, sc.w, sh.w);
                    if (plain) {
                        const float4 r = cur_a;
                        if (kRes != 0 && kRes != 2 && res_first) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                        if (act_ == FIERY_ACT_RELU) {
                            v.x = fmaxf(v.x, 0.f);  v.y = fmaxf(v.y, 0.f);  v.z = fmaxf(v.z, 0.f);  v.w = fmaxf(v.w, 0.f);
                        } else if (act_ == FIERY_ACT_SIGMOID) {
                            v.x = sigmoidf(v.x);  v.y = sigmoidf(v.y);  v.z = sigmoidf(v.z);  v.w = sigmoidf(v.w);
                        } else if (act_ == FIERY_ACT_SWISH) {
                            v.x *= sigmoid_swish(v.x);  v.y *= sigmoid_swish(v.y);  v.z *= sigmoid_swish(v.z);  v.w *= sigmoid_swish(v.w);
                        }
                        if (kRes != 0 && kRes != 1 && !res_first) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                        store4(r_out, o_out, v);
                    } else if (gates) {
                        float4 g = make_float4(sigmoid_gate(v.x), sigmoid_gate(v.y), sigmoid_gate(v.z), sigmoid_gate(v.w));
                        if (upper) {                                                                // (1 - reset) * state
                            const float4 h = cur_a;
                            g.x = (1.0f - g.x) * h.x;  g.y = (1.0f - g.y) * h.y;  g.z = (1.0f - g.z) * h.z;  g.w = (1.0f - g.w) * h.w;
                            store4(r_out2, o_out2, g);
                        } else {
                            store4(r_out, o_out, g);                                                // update gate
                        }
                    } else {                                                                        // FIERY_EPI_GRU_OUT
                        const float4 u = cur_a, h = cur_b;
                        float4 hn;
                        { const float a = (1.0f - u.x) * h.x, b = u.x * fmaxf(v.x, 0.f); hn.x = a + b; }
                        { const float a = (1.0f - u.y) * h.y, b = u.y * fmaxf(v.y, 0.f); hn.y = a + b; }
                        { const float a = (1.0f - u.z) * h.z, b = u.z * fmaxf(v.z, 0.f); hn.z = a + b; }
                        { const float a = (1.0f - u.w) * h.w, b = u.w * fmaxf(v.w, 0.f); hn.w = a + b; }
                        store4(r_out, o_out, hn);
                        if (has_d1) store4(r_out2, o_out2, hn);
                    }
                    cur_a = nxt_a;  cur_b = nxt_b;  cur_bias = nxt_bias;
                    gp = gp_n;
                    o_out += s_out;  o_out2 += s_out2;  o_a += s_a;  o_b += s_b;
                }
            };
            using std::integral_constant;
            constexpr integral_constant<int, -1> any{};
            constexpr integral_constant<int, 0> c0{};
            constexpr integral_constant<int, 1> c1{};
            constexpr integral_constant<int, 2> c2{};
            const bool no_bias = !(with_bias && p.img_bias);
            if (no_bias && p.epi == FIERY_EPI_PLAIN && act == FIERY_ACT_RELU) {
                constexpr integral_constant<int, FIERY_EPI_PLAIN> e{};
                constexpr integral_constant<int, FIERY_ACT_RELU> a{};
                if (!p.res.ptr) rows(e, a, c0, c0);
                else if (res_pre_rows) rows(e, a, c1, c0);
                else rows(e, a, c2, c0);
            } else if (no_bias && p.epi == FIERY_EPI_PLAIN && act == FIERY_ACT_NONE && !p.res.ptr) {
                rows(integral_constant<int, FIERY_EPI_PLAIN>{}, integral_constant<int, FIERY_ACT_NONE>{}, c0, c0);
            } else if (no_bias && p.epi == FIERY_EPI_GRU_GATES) {
                rows(integral_constant<int, FIERY_EPI_GRU_GATES>{}, any, c0, c0);
            } else if (no_bias && p.epi == FIERY_EPI_GRU_OUT) {
                rows(integral_constant<int, FIERY_EPI_GRU_OUT>{}, any, c0, c0);
            } else {
                rows(any, any, any, any);
            }
            return;
        }
        int gp = pix0 + prow0;
        int o = fast_div(gp, p.mg_hw, p.sh_hw), ppi = gp - o * HWout;
        // The row's global operands - residual, or the GRU's state / update gate, and the per-image bias - are requested
        // one row AHEAD of their use: destination and operands are not known to be distinct tensors, so the compiler keeps
        // a row's loads behind the previous row's store, i.e. one exposed memory round trip per row (8 to 16 rows per
        // thread; measured as ~11 k cycles of a 34 k-cycle chained epilogue).  Fetching row i + 1 before storing row i is
        // correct even for an in-place residual: the rows are different pixels.
        struct RowOperands {
            float4 a, b, bias;
        };
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto fetch = [&](int o_, int ppi_, bool live) {
            RowOperands r{zero4, zero4, zero4};
            if (!live) return r;
            const long long pp = ppi_;
            if (with_bias && p.img_bias) {
                long long brow = o_;
                if (p.bias_border) {                        // 3x3 class of (y, x): which taps fall inside the image
                    const int y = fast_div(ppi_, p.mg_w, p.sh_w), x = ppi_ - y * p.Wout;
                    brow = brow * 9 + (y == 0 ? 0 : y == p.Hout - 1 ? 2 : 1) * 3 + (x == 0 ? 0 : x == p.Wout - 1 ? 2 : 1);
                }
                r.bias = *reinterpret_cast<const float4*>(p.img_bias + brow * p.cout_pad + co);
            }
            if (p.epi == FIERY_EPI_PLAIN) {
                if (p.res.ptr) r.a = *reinterpret_cast<const float4*>(p.res.ptr + o_ * p.res.istride + pp * p.res.ld + co);
            } else if (p.epi == FIERY_EPI_GRU_GATES) {
                if (gates_upper) r.a = *reinterpret_cast<const float4*>(p.aux0.ptr + o_ * p.aux0.istride + pp * p.aux0.ld + (co - half));
            } else {                                                                        // FIERY_EPI_GRU_OUT
                r.a = *reinterpret_cast<const float4*>(p.aux0.ptr + o_ * p.aux0.istride + pp * p.aux0.ld + co);
                r.b = *reinterpret_cast<const float4*>(p.aux1.ptr + o_ * p.aux1.istride + pp * p.aux1.ld + co);
            }
            return r;
        };
        RowOperands cur = fetch(o, ppi, gp < M);
        for (int pl = prow0; pl < BM; pl += rows_per_pass) {
            if (gp >= M) break;
            // next row of this thread: its pixel, and its operands on their way
            int gp_n = gp + rows_per_pass, ppi_n = ppi + rows_per_pass, o_n = o;
            while (ppi_n >= HWout) {
                ppi_n -= HWout;
                ++o_n;
            }
            const RowOperands nxt = fetch(o_n, ppi_n, pl + rows_per_pass < BM && gp_n < M);
            float4 v = *reinterpret_cast<const float4*>(&smem[pl * width + c4 * 4]);
            v.x += cur.bias.x;  v.y += cur.bias.y;  v.z += cur.bias.z;  v.w += cur.bias.w;
            v.x = fmaf(v.x, sc.x, sh.x);  v.y = fmaf(v.y, sc.y, sh.y);  v.z = fmaf(v.z, sc.z, sh.z);  v.w = fmaf(v.w, sc.w, sh.w);
            const long long pp = ppi;
            if (p.epi == FIERY_EPI_PLAIN) {
                const float4 r = cur.a;
                if (res_pre_rows) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                if (act == FIERY_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f);  v.y = fmaxf(v.y, 0.f);  v.z = fmaxf(v.z, 0.f);  v.w = fmaxf(v.w, 0.f);
                } else if (act == FIERY_ACT_SIGMOID) {
                    v.x = sigmoidf(v.x);  v.y = sigmoidf(v.y);  v.z = sigmoidf(v.z);  v.w = sigmoidf(v.w);
                } else if (act == FIERY_ACT_SWISH) {
                    v.x *= sigmoid_swish(v.x);  v.y *= sigmoid_swish(v.y);  v.z *= sigmoid_swish(v.z);  v.w *= sigmoid_swish(v.w);
                }
                if (!res_pre_rows) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                *reinterpret_cast<float4*>(p.out.ptr + o * p.out.istride + pp * p.out.ld + co) = v;
            } else if (p.epi == FIERY_EPI_GRU_GATES) {
                float4 g = make_float4(sigmoid_gate(v.x), sigmoid_gate(v.y), sigmoid_gate(v.z), sigmoid_gate(v.w));
                if (!gates_upper) {                                                         // update gate
                    *reinterpret_cast<float4*>(p.out.ptr + o * p.out.istride + pp * p.out.ld + co) = g;
                } else {                                                                    // (1 - reset) * state
                    const int c2 = co - half;
                    const float4 h = cur.a;
                    g.x = (1.0f - g.x) * h.x;  g.y = (1.0f - g.y) * h.y;  g.z = (1.0f - g.z) * h.z;  g.w = (1.0f - g.w) * h.w;
                    *reinterpret_cast<float4*>(p.out2.ptr + o * p.out2.istride + pp * p.out2.ld + c2) = g;
                }
            } else {                                                                        // FIERY_EPI_GRU_OUT
                const float4 u = cur.a, h = cur.b;
                float4 hn;
                { const float a = (1.0f - u.x) * h.x, b = u.x * fmaxf(v.x, 0.f); hn.x = a + b; }
                { const float a = (1.0f - u.y) * h.y, b = u.y * fmaxf(v.y, 0.f); hn.y = a + b; }
                { const float a = (1.0f - u.z) * h.z, b = u.z * fmaxf(v.z, 0.f); hn.z = a + b; }
                { const float a = (1.0f - u.w) * h.w, b = u.w * fmaxf(v.w, 0.f); hn.w = a + b; }
                *reinterpret_cast<float4*>(p.out.ptr + o * p.out.istride + pp * p.out.ld + co) = hn;
                if (p.out2.ptr) *reinterpret_cast<float4*>(p.out2.ptr + o * p.out2.istride + pp * p.out2.ld + co) = hn;
            }
            cur = nxt;
            gp = gp_n;
            ppi = ppi_n;
            o = o_n;
        }
    };
    auto stage_tile = [&](const v16f& a, int t, int nt, int width) {
        const int col = wn * (32 * NT) + nt * 32 + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pl = wm * (32 * MT) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            smem[pl * width + col] = a[r];
        }
    };
    const bool rows16 = (p.vec_epilogue & 1) != 0;

    // ---- Bottleneck tail (CHAIN kernels): 3 x 3 -> BN + act -> 1 x 1 (32 -> 64) -> BN + act + residual [-> 1 x 1 (64 -> 32) -> BN + act]
    // (layers/convolutions.py:110-168 of the reference).  Lane = pixel, registers = couts (PIXLANE above): the chained products
    // take their input straight from the accumulator registers, and a wavefront finishes its own 32 pixels without meeting the
    // other three - no workgroup barrier after the K loop.  Global rows are still moved as FULL rows (a lane's own 16-byte
    // pieces of its pixel row would be 32-byte fragments of 32 different cache lines per instruction - measured 16 % slower
    // than the staged rows, profiles/r5_tail_ab.txt): the accumulators pass through an exchange tile in LDS that belongs to
    // the wavefront alone, ordered by `wave_sync` (the LDS runs one wavefront's instructions in order).
    if constexpr (PIXLANE) {
        auto comp = [](const float4& f, int j) { return j == 0 ? f.x : j == 1 ? f.y : j == 2 ? f.z : f.w; };
        auto activate = [](float v, int act) {
            if (act == FIERY_ACT_RELU) return fmaxf(v, 0.f);
            if (act == FIERY_ACT_SIGMOID) return sigmoidf(v);
            if (act == FIERY_ACT_SWISH) return v * sigmoid_swish(v);
            return v;
        };
        auto ld4 = [&](const float* q, bool vec) {                   // four consecutive floats; 16-byte access when allowed
            if (vec) return *reinterpret_cast<const float4*>(q);
            return make_float4(q[0], q[1], q[2], q[3]);
        };
        // the second product's weights are requested first - the packed image [k / 4][cout][k % 4] IS the order the accumulator
        // registers hold their couts in: k = c(4 g + j, hi) sits at piece 2 g + hi, element j
        // (SPLIT: p.w2 is the split image [term][k half][cout block][lane] of 16-byte operands - fiery_conv_desc.weights2_split)
        float4 w2r[SPLIT ? 12 : 8];
        if constexpr (SPLIT) {
#pragma unroll
            for (int q = 0; q < 12; ++q) w2r[q] = reinterpret_cast<const float4*>(p.w2)[q * 64 + lane];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) w2r[g * 2 + nt] = reinterpret_cast<const float4*>(p.w2)[(2 * g + hi) * 64 + nt * 32 + m];
        }
        const int wpix0 = pix0 + wm * 32;                            // this wavefront's 32 pixels
        // where a pixel's row starts in a tensor (floats from its base): dense tensors - images back to back - need one multiply
        const bool dense = (p.vec_epilogue & 2) != 0;
        auto row_at = [&](const TensP& t, int gp) -> long long {
            if (dense) return static_cast<long long>(gp) * t.ld;
            const int o = fast_div(gp, p.mg_hw, p.sh_hw);
            return o * t.istride + static_cast<long long>(gp - o * HWout) * t.ld;
        };
        // row layout of the 64-cout rows: sixteen 16-byte chunks per row, four rows per pass, eight passes
        const int c16 = lane & 15, r16 = lane >> 4;
        float4 resid[8];
        if (rows16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gp = wpix0 + 4 * i + r16;
                resid[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.res.ptr && gp < M && 4 * c16 < p.cout_store) resid[i] = *reinterpret_cast<const float4*>(p.res.ptr + row_at(p.res, gp) + 4 * c16);
            }
        }
        // (1) h = act(acc * scale + shift): register r of this lane half is cout c(r, hi)
        float h[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 sc = ld4(p.scale + 8 * g + 4 * hi, rows16), sh = ld4(p.shift + 8 * g + 4 * hi, rows16);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[4 * g + j] = activate(fmaf(acc[0][4 * g + j], comp(sc, j), comp(sh, j)), p.act);
        }
        // (2) out2[cout2][pixel] += W2[k][cout2] h[pixel][k]: weights as A (row = cout2), h as B (column = pixel) - the result is
        //     lane = pixel, registers = couts again
        v16f acc2[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
        if constexpr (SPLIT) {
            // three-term operands, six partial products (smallest first), like the K loop: h is split here, W2 arrived split
            constexpr int TW[6] = {0, 2, 1, 0, 1, 0}, TH[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float x[8] = {h[8 * i], h[8 * i + 1], h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6], h[8 * i + 7]};
                bf16x8 hb[3];
                split_bf16x8(x, hb[0], hb[1], hb[2]);
#pragma unroll
                for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc2[nt] = mfma_bf16_32x32x16(bits_bf16x8(w2r[(TW[k6] * 2 + i) * 2 + nt]), hb[TH[k6]], acc2[nt]);
            }
        } else if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 hb = pack_bf16x8(make_float4(h[8 * i], h[8 * i + 1], h[8 * i + 2], h[8 * i + 3]),
                                              make_float4(h[8 * i + 4], h[8 * i + 5], h[8 * i + 6], h[8 * i + 7]));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc2[nt] = mfma_bf16_32x32x16(pack_bf16x8(w2r[(2 * i) * 2 + nt], w2r[(2 * i + 1) * 2 + nt]), hb, acc2[nt]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(w2r[(r >> 2) * 2 + nt], r & 3), h[r], acc2[nt], 0, 0, 0);
        }
        if (!rows16) {
            // unaligned destinations: this lane's pixel, channel by channel, straight from the registers (no third stage here)
            const int gp = wpix0 + m;
            if (gp < M) {
                const long long at_out = row_at(p.out, gp), at_res = p.res.ptr ? row_at(p.res, gp) : 0;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = nt * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                        if (co >= p.cout_store) continue;
                        float v = activate(fmaf(acc2[nt][r], p.scale2[co], p.shift2[co]), p.act2);
                        if (p.res.ptr) v += p.res.ptr[at_res + co];
                        p.out.ptr[at_out + co] = v;
                    }
            }
            clk_finish();
            return;
        }
        // the third product's weights (the NEXT block's 1 x 1 down-projection, 64 -> 32): packed [k / 32][k % 32 / 4][cout][k % 4]
        const bool third = p.heads.w != nullptr;
        // (SPLIT: p.heads.w is the split image [term][k block of 32][half][lane] - fiery_conv_desc.weights3_split)
        float4 w3r[SPLIT ? 12 : 8];
        if constexpr (SPLIT) {
#pragma unroll
            for (int q = 0; q < 12; ++q) w3r[q] = third ? reinterpret_cast<const float4*>(p.heads.w)[q * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                w3r[i] = third ? reinterpret_cast<const float4*>(p.heads.w)[(i >> 2) * 256 + (2 * (i & 3) + hi) * 32 + m] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // (3) the accumulators through the wavefront's exchange tile into rows: BN + activation + residual (after the
        //     activation) on full rows - a lane's four channels are the same for all its rows, so scale / shift are two loads
        float4* const xt = reinterpret_cast<float4*>(smem + wv * (32 * CHAIN_PITCH));       // [32 rows][17 x 16 bytes]
        constexpr int XP = CHAIN_PITCH / 4;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xt[m * XP + nt * 8 + 2 * g + hi] = make_float4(acc2[nt][4 * g], acc2[nt][4 * g + 1], acc2[nt][4 * g + 2], acc2[nt][4 * g + 3]);
        wave_sync();
        {
            const int co = 4 * c16;
            const float4 sc = *reinterpret_cast<const float4*>(p.scale2 + co), sh = *reinterpret_cast<const float4*>(p.shift2 + co);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + r16, gp = wpix0 + row;
                float4 v = xt[row * XP + c16];
                v.x = activate(fmaf(v.x, sc.x, sh.x), p.act2) + resid[i].x;
                v.y = activate(fmaf(v.y, sc.y, sh.y), p.act2) + resid[i].y;
                v.z = activate(fmaf(v.z, sc.z, sh.z), p.act2) + resid[i].z;
                v.w = activate(fmaf(v.w, sc.w, sh.w), p.act2) + resid[i].w;
                const bool keep = gp < M && co < p.cout_store;
                if (keep) *reinterpret_cast<float4*>(p.out.ptr + row_at(p.out, gp) + co) = v;
                if (third) xt[row * XP + c16] = keep ? v : make_float4(0.f, 0.f, 0.f, 0.f);     // (rows that do not exist: zeros)
            }
        }
        if (third) {
            // (4) ... and the next Bottleneck's 1 x 1 down-projection (64 -> 32, + BN + ReLU) on the finished rows.  On its own that
            //     layer is a memory-bound launch that re-reads what this kernel has just written.  Operands ride in members the
            //     chained mode does not use: heads.w = packed 64 x 32 weights, aux0.ptr / aux1.ptr = scale3 / shift3 [32],
            //     heads.n_out = act3, out2 = destination.
            wave_sync();
            v16f acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float4 y4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) y4[g] = xt[m * XP + nt * 8 + 2 * g + hi];
                if constexpr (SPLIT) {
                    constexpr int TW[6] = {0, 2, 1, 0, 1, 0}, TY[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float x[8] = {y4[2 * i].x, y4[2 * i].y, y4[2 * i].z, y4[2 * i].w, y4[2 * i + 1].x, y4[2 * i + 1].y, y4[2 * i + 1].z, y4[2 * i + 1].w};
                        bf16x8 yb[3];
                        split_bf16x8(x, yb[0], yb[1], yb[2]);
#pragma unroll
                        for (int k6 = 0; k6 < 6; ++k6)
                            acc3 = mfma_bf16_32x32x16(bits_bf16x8(w3r[(TW[k6] * 2 + nt) * 2 + i]), yb[TY[k6]], acc3);
                    }
                } else if constexpr (BF16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc3 = mfma_bf16_32x32x16(pack_bf16x8(w3r[nt * 4 + 2 * i], w3r[nt * 4 + 2 * i + 1]), pack_bf16x8(y4[2 * i], y4[2 * i + 1]), acc3);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(w3r[nt * 4 + (r >> 2)], r & 3), comp(y4[r >> 2], r & 3), acc3, 0, 0, 0);
                }
            }
            wave_sync();                                           // every lane has read its operands: the tile takes the result
#pragma unroll
            for (int g = 0; g < 4; ++g) xt[m * XP + 2 * g + hi] = make_float4(acc3[4 * g], acc3[4 * g + 1], acc3[4 * g + 2], acc3[4 * g + 3]);
            wave_sync();
            // rows of 32 couts: eight 16-byte chunks per row, eight rows per pass, four passes
            const int c8 = lane & 7, r8 = lane >> 3;
            const int dco = 4 * c8;
            const float4 sc = *reinterpret_cast<const float4*>(p.aux0.ptr + dco), sh = *reinterpret_cast<const float4*>(p.aux1.ptr + dco);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + r8, gp = wpix0 + row;
                float4 v = xt[row * XP + c8];
                v.x = fmaf(v.x, sc.x, sh.x);  v.y = fmaf(v.y, sc.y, sh.y);  v.z = fmaf(v.z, sc.z, sh.z);  v.w = fmaf(v.w, sc.w, sh.w);
                if (p.heads.n_out == FIERY_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f);  v.y = fmaxf(v.y, 0.f);  v.z = fmaxf(v.z, 0.f);  v.w = fmaxf(v.w, 0.f);
                }
                if (gp < M) *reinterpret_cast<float4*>(p.out2.ptr + row_at(p.out2, gp) + dco) = v;
            }
        }
        clk_finish();
        return;
    }

    // ---- decoder heads: the hidden tile stays in LDS, only the heads' final 1x1 outputs are stored -------------
    if constexpr (BM == 64 && BN == 128) {
        if (p.epi == FIERY_EPI_HEADS) {
            float* w2 = smem + BM * BN;                              // [<= 4 outputs of this cout tile][64], behind the tile
            if constexpr (NT == 2) {
                stage_tile(acc[0], 0, 0, BN);
                stage_tile(acc[1], 0, 1, BN);
            } else {                                                 // (halo loop: two pixel tiles, one cout tile per wavefront)
                stage_tile(acc[0], 0, 0, BN);
                stage_tile(acc[1], 1, 0, BN);
            }
            // the final 1x1 rows whose 64-channel group lies in this 128-cout tile (at most 4: one per wavefront)
            int my_out = -1;
            {
                int slot = 0;
                for (int o = 0; o < p.heads.n_out; ++o) {
                    if ((p.heads.group[o] >> 1) != tile_n) continue;
                    if (slot < 4 && tid < 64) w2[slot * 64 + tid] = p.heads.w[o * 64 + tid];
                    if (slot == wv) my_out = o;
                    ++slot;
                }
            }
            __syncthreads();
            // hidden = act(acc * scale + shift), in place, 16 bytes per thread and pass
            {
                const int c4 = tid & 31, prow0 = tid >> 5;            // 32 chunks of four couts, 8 rows per pass
                const float4 sc = *reinterpret_cast<const float4*>(p.scale + tile_n * BN + c4 * 4);
                const float4 sh = *reinterpret_cast<const float4*>(p.shift + tile_n * BN + c4 * 4);
                for (int pl = prow0; pl < BM; pl += 8) {
                    float4 v = *reinterpret_cast<float4*>(&smem[pl * BN + c4 * 4]);
                    v.x = fmaf(v.x, sc.x, sh.x);  v.y = fmaf(v.y, sc.y, sh.y);  v.z = fmaf(v.z, sc.z, sh.z);  v.w = fmaf(v.w, sc.w, sh.w);
                    if (p.act == FIERY_ACT_RELU) {
                        v.x = fmaxf(v.x, 0.f);  v.y = fmaxf(v.y, 0.f);  v.z = fmaxf(v.z, 0.f);  v.w = fmaxf(v.w, 0.f);
                    } else if (p.act == FIERY_ACT_SIGMOID) {
                        v.x = sigmoidf(v.x);  v.y = sigmoidf(v.y);  v.z = sigmoidf(v.z);  v.w = sigmoidf(v.w);
                    }
                    *reinterpret_cast<float4*>(&smem[pl * BN + c4 * 4]) = v;
                }
            }
            __syncthreads();
            // wavefront wv owns one output row; a lane owns a pixel.  Lane l starts at channel l and walks round, so the
            // 64 lanes touch 32 different banks at every step although their rows are 512 bytes apart.
            if (my_out >= 0) {
                const int slot = wv;                                   // == position of my_out among this tile's rows
                const int cb = (p.heads.group[my_out] & 1) * 64;
                const float* row = &smem[lane * BN + cb];
                const float* wrow = &w2[slot * 64];
                float a = 0.f;
#pragma unroll 8
                for (int c0 = 0; c0 < 64; ++c0) {
                    const int cc = (c0 + lane) & 63;
                    a = fmaf(row[cc], wrow[cc], a);
                }
                a += p.heads.bias[my_out];
                if (p.heads.sigmoid[my_out]) a = sigmoidf(a);
                const int gp = pix0 + lane;
                if (gp < M) {
                    const int o = fast_div(gp, p.mg_hw, p.sh_hw), ppi = gp - o * HWout;
                    p.heads.out[my_out][o * p.heads.istride[my_out] + ppi] = a;
                }
            }
            return;
        }
    }

    if (rows16) {
        // (the loop's last barrier has passed: the stages are free)
        float4 fetched[2];
        {
            const int co = tile_n * BN + (tid % (BN >> 2)) * 4;
            const bool live = co < p.cout_pad;
            fetched[0] = live ? *reinterpret_cast<const float4*>(p.scale + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            fetched[1] = live ? *reinterpret_cast<const float4*>(p.shift + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stage_tile(acc[0], 0, 0, BN);
        if constexpr (NT == 2) stage_tile(acc[1], 0, 1, BN);
        if constexpr (MT >= 2) {
            stage_tile(acc[NT], 1, 0, BN);
            if constexpr (NT == 2) stage_tile(acc[NT + 1], 1, 1, BN);
        }
        if constexpr (MT == 4) {                                   // (halo loop, 128 x 128: one wavefront holds all four pixel tiles)
            stage_tile(acc[2 * NT], 2, 0, BN);
            stage_tile(acc[3 * NT], 3, 0, BN);
        }
        __syncthreads();
        if constexpr (CLK) {
            if (clk_trace) clk_trace[6] = wall_clock64();                  // tile staged, every wavefront through the barrier
        }
        store_rows(BN, tile_n * BN, p.scale, p.shift, p.act, true, fetched);
        clk_finish();
        return;
    }

    // ---- register epilogue (unaligned destinations): a lane holds one cout for 16 pixel rows ---------------------
    const int half = p.cout_pad >> 1;
    auto emit = [&](const v16f& a, int t, int nt) {
        const int co = tile_n * BN + wn * (32 * NT) + nt * 32 + m;
        const float sc = p.scale[co], sh = p.shift[co];
        // image / in-image pixel of this lane's first row; the other 15 rows are small constant offsets away,
        // so one division per tile instead of one per element
        const int gp_base = pix0 + wm * (32 * MT) + t * 32 + 4 * hi;
        const int o_base = fast_div(gp_base, p.mg_hw, p.sh_hw);
        const int pp_base = gp_base - o_base * HWout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = (r & 3) + 8 * (r >> 2);
            if (gp_base + off >= M) continue;
            int o = o_base, ppi = pp_base + off;
            while (ppi >= HWout) {
                ppi -= HWout;
                ++o;
            }
            const long long pp = ppi;
            float v = a[r];
            if (p.img_bias) {
                long long brow = o;
                if (p.bias_border) {
                    const int y = fast_div(ppi, p.mg_w, p.sh_w), x = ppi - y * p.Wout;
                    brow = brow * 9 + (y == 0 ? 0 : y == p.Hout - 1 ? 2 : 1) * 3 + (x == 0 ? 0 : x == p.Wout - 1 ? 2 : 1);
                }
                v += p.img_bias[brow * p.cout_pad + co];
            }
            v = fmaf(v, sc, sh);
            if (p.epi == FIERY_EPI_PLAIN) {
                if (co >= p.cout_store) continue;
                if (p.res.ptr && p.res_pre) v += p.res.ptr[o * p.res.istride + pp * p.res.ld + co];
                if (p.act == FIERY_ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == FIERY_ACT_SIGMOID) v = sigmoidf(v);
                else if (p.act == FIERY_ACT_SWISH) v *= sigmoid_swish(v);
                if (p.res.ptr && !p.res_pre) v += p.res.ptr[o * p.res.istride + pp * p.res.ld + co];
                p.out.ptr[o * p.out.istride + pp * p.out.ld + co] = v;
            } else if (p.epi == FIERY_EPI_GRU_GATES) {
                const float g = sigmoidf(v);
                if ((co < half ? co : co - half) >= p.cout_store) continue;
                if (co < half) {
                    p.out.ptr[o * p.out.istride + pp * p.out.ld + co] = g;                       // update gate
                } else {
                    const int c2 = co - half;
                    const float h = p.aux0.ptr[o * p.aux0.istride + pp * p.aux0.ld + c2];
                    p.out2.ptr[o * p.out2.istride + pp * p.out2.ld + c2] = (1.0f - g) * h;       // (1 - reset) * state
                }
            } else {   // FIERY_EPI_GRU_OUT
                if (co >= p.cout_store) continue;
                const float ht = fmaxf(v, 0.f);
                const float u = p.aux0.ptr[o * p.aux0.istride + pp * p.aux0.ld + co];
                const float h = p.aux1.ptr[o * p.aux1.istride + pp * p.aux1.ld + co];
                const float x1 = (1.0f - u) * h;
                const float x2 = u * ht;
                const float hn = x1 + x2;
                p.out.ptr[o * p.out.istride + pp * p.out.ld + co] = hn;
                if (p.out2.ptr) p.out2.ptr[o * p.out2.istride + pp * p.out2.ld + co] = hn;
            }
        }
    };
    // compile-time tile indices keep the accumulators in registers
    emit(acc[0], 0, 0);
    if constexpr (NT == 2) emit(acc[1], 0, 1);
    if constexpr (MT >= 2) {
        emit(acc[NT], 1, 0);
        if constexpr (NT == 2) emit(acc[NT + 1], 1, 1);
    }
    if constexpr (MT == 4) {
        emit(acc[2 * NT], 2, 0);
        emit(acc[3 * NT], 3, 0);
    }
}

// A workgroup takes the pixel tiles bid, bid + gridDim.x, ... one after another.  By default the grid has one workgroup per
// tile and the loop runs once; FIERY_CONV_PERSISTENT=1 caps the grid at one workgroup per slot of the chip (round 4's
// experiment: would workgroups that stay hide each other's epilogue and set-up under their K loops?  see conv_persistent_grid).
template <int BM, int BN, bool CLK = false, int PRIO = 0, bool SMALLCIN = false, bool ALIGNED = false, bool BF16 = false, bool HALO = false,
          bool CHAIN = false, bool SK = false, bool SPLIT = false>
__global__ __launch_bounds__(256, conv_waves_per_simd(BM, BN, ALIGNED && !CLK && !BF16, BF16, HALO, CHAIN, SPLIT)) void k_conv_igemm(ConvP p) {
    if constexpr (SK) {
        // STREAM-K (round 5): the launch's work - every (output tile, K chunk) unit - is dealt out EVENLY to exactly as many
        // workgroups as the chip holds at once; a workgroup walks its contiguous run of units tile by tile, so a tile's chunks
        // may be shared by two or more workgroups (conv_tile: partial tiles).  No partly filled last round of workgroups: that
        // quantisation, not the loop, was what held the big layers at 0.72-0.76 of the matrix peak (938 tiles on 512 slots).
        // Workgroup b runs on XCD b % 8: the runs are dealt so that each XCD gets a contiguous eighth of the unit space and
        // neighbouring tiles share their halo rows through that XCD's L2.
        const int nwg = static_cast<int>(gridDim.x), b = static_cast<int>(blockIdx.x);
        const int j = (b & 7) * (nwg >> 3) + (b >> 3);
        long long u0, u1;
        {
            const ConvP& q = kernel_args_again(p);
            const long long total = static_cast<long long>(q.sk_tiles) * q.k_chunks;
            u0 = j * total / nwg;
            u1 = (j + 1) * total / nwg;
        }
        while (u0 < u1) {
            const ConvP& q = kernel_args_again(p);
            const int kc = q.k_chunks;
            const int tile = static_cast<int>(u0 / kc);
            const int kb = static_cast<int>(u0 - static_cast<long long>(tile) * kc);
            const long long left = u1 - u0;
            const int ke = left < kc - kb ? kb + static_cast<int>(left) : kc;
            conv_tile<BM, BN, CLK, PRIO, SMALLCIN, ALIGNED, BF16, HALO, CHAIN, true>(q, 0, 0, tile, kb, ke, j, nwg);
            u0 += ke - kb;
            if (u0 < u1) __syncthreads();                      // the next tile's first stage overwrites the staging tile
        }
        return;
    }
    const int n_tiles_m = p.tiles_m;
    for (int bid = blockIdx.x; bid < n_tiles_m; bid += gridDim.x) {
        // the argument block is read afresh for every tile (kernel_args_again: an offset the optimiser cannot see through):
        // hoisted out of the loop its hundred scalars stay live across the tile and spill (168 registers + 496 B of scratch);
        // read in place, never copied: the heads' members are indexed at run time and a copy would live in scratch
        conv_tile<BM, BN, CLK, PRIO, SMALLCIN, ALIGNED, BF16, HALO, CHAIN, false, SPLIT>(kernel_args_again(p), bid, n_tiles_m);
        if (bid + static_cast<int>(gridDim.x) < n_tiles_m) __syncthreads();      // the next tile's first stage overwrites the staging tile
    }
}

// The grid of a launch: one workgroup per pixel tile and cout tile, capped at the workgroups the chip holds at once
// (`per_cu` per CU - the kernel's launch bound; a multiple of 8 in x so that the tile order stays XCD-aware); the kernel's
// workgroups then take several pixel tiles each.  FIERY_CONV_PERSISTENT=0: a workgroup per tile (A/B runs).
inline dim3 conv_persistent_grid(ConvP& q, dim3 grid, int per_cu) {
    q.tiles_m = static_cast<int>(grid.x);
    // (the switch is read once per process and the launch returns at once when it is off: this runs for every launch)
    static const bool persistent = [] {
        const char* e = getenv("FIERY_CONV_PERSISTENT");
        return e && atoi(e) != 0;
    }();
    if (!persistent) return grid;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    // Measured (profiles/r4_conv_phases.txt, r4_conv_persistent_ab.txt): NOT faster - 281 against 286 samples/s on baseline.yml;
    // the workgroups of a CU are not in step to begin with (a tile's K loop takes 55-115 us depending on what its neighbours
    // are doing), so there was no idle phase to fill, and a persistent workgroup's set-up meets two running K loops.  Opt-in.
    int cap = (n_cu * per_cu / static_cast<int>(grid.y > 0 ? grid.y : 1)) & ~7;
    if (cap < 8) cap = 8;
    if (static_cast<int>(grid.x) > cap) grid.x = static_cast<unsigned>(cap);
    return grid;
}
#define FIERY_CONV_LAUNCH_C(BM_, BN_, CLK_, PRIO_, SMALL_, ALIGNED_, BF16_, HALO_, CHAIN_)                                    \
    do {                                                                                                                     \
        ConvP q_ = p;                                                                                                        \
        const dim3 g_ = conv_persistent_grid(q_, grid, conv_waves_per_simd(BM_, BN_, ALIGNED_ && !CLK_ && !BF16_, BF16_, HALO_, CHAIN_)); \
        hipLaunchKernelGGL((k_conv_igemm<BM_, BN_, CLK_, PRIO_, SMALL_, ALIGNED_, BF16_, HALO_, CHAIN_>), g_, dim3(256), 0, hs, q_); \
    } while (0)
#define FIERY_CONV_LAUNCH(BM_, BN_, CLK_, PRIO_, SMALL_, ALIGNED_, BF16_, HALO_)                                              \
    FIERY_CONV_LAUNCH_C(BM_, BN_, CLK_, PRIO_, SMALL_, ALIGNED_, BF16_, HALO_, false)

// ---- weight packing ------------------------------------------------------------------------------
// The variants of tile shape (BM, BN) that exist: generic and scalar-addressed for all five shapes, the small-cin
// loop for the four shapes that use it, the probes only in tuning builds (FIERY_CONV_TUNING=1: they double the
// compile time and nothing in the product path launches them).
// kMask: which variants this translation unit instantiates (bit = ConvVariant); the 128 x 128 tile, whose kernels take
// minutes each to compile, is spread over two units.
template <int BM, int BN>
void conv_launch_tile_bf16(const ConvP& p, dim3 grid, hipStream_t hs) {
    if constexpr (BM == 128 && BN == 32) {
        if (p.w2) {                                          // Bottleneck tail: the register-chained kernel
            FIERY_CONV_LAUNCH_C(BM, BN, false, 0, false, true, true, false, true);
            return;
        }
    }
    FIERY_CONV_LAUNCH(BM, BN, false, 0, false, true, true, false);
}
// the split form (fp32 accuracy on the bf16 matrix cores; weights packed by fiery_conv_pack_weights_split in p.w)
template <int BM, int BN>
void conv_launch_tile_split(const ConvP& p, dim3 grid, hipStream_t hs) {
    ConvP q_ = p;
    const dim3 g_ = conv_persistent_grid(q_, grid, conv_waves_per_simd(BM, BN, false, true, false, false, true));
    if constexpr (BM == 128 && BN == 32) {
        if (p.w2) {                                          // Bottleneck tail: the register-chained kernel
            hipLaunchKernelGGL((k_conv_igemm<BM, BN, false, 0, false, true, true, false, true, false, true>), g_, dim3(256), 0, hs, q_);
            return;
        }
    }
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, false, 0, false, true, true, false, false, false, true>), g_, dim3(256), 0, hs, q_);
}
template <int BM, int BN>
void conv_launch_tile_stream_k(const ConvP& p, dim3 grid, hipStream_t hs) {
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, false, 0, false, true, false, false, false, true>), grid, dim3(256), 0, hs, p);
}
template <int BM, int BN>
void conv_launch_tile_f32_halo(const ConvP& p, dim3 grid, hipStream_t hs) {
    FIERY_CONV_LAUNCH(BM, BN, false, 0, false, true, false, true);
}
template <int BM, int BN>
void conv_launch_tile_bf16_halo(const ConvP& p, dim3 grid, hipStream_t hs) {
    FIERY_CONV_LAUNCH(BM, BN, false, 0, false, true, true, true);
}

template <int BM, int BN, unsigned kMask>
bool conv_launch_tile(const ConvP& p, dim3 grid, hipStream_t hs, int variant, unsigned long long* clk) {
    (void)clk;
    if constexpr (BM == 128 && BN == 32) {
        // Bottleneck tails (a chained 1 x 1): the register-chained kernels - every loop variant has one, the probes none
        if (p.w2) {
            if (variant == kConvAligned || variant == kConvClockAligned) FIERY_CONV_LAUNCH_C(BM, BN, false, 0, false, true, false, false, true);
            else if (variant == kConvSmallCin) FIERY_CONV_LAUNCH_C(BM, BN, false, 0, true, false, false, false, true);
            else FIERY_CONV_LAUNCH_C(BM, BN, false, 0, false, false, false, false, true);
            return true;
        }
    }
    if constexpr ((kMask >> kConvGeneric) & 1u) {
        if (variant == kConvGeneric) {
            FIERY_CONV_LAUNCH(BM, BN, false, 0, false, false, false, false);
            return true;
        }
    }
    if constexpr ((kMask >> kConvAligned) & 1u) {
        if (variant == kConvAligned) {
            FIERY_CONV_LAUNCH(BM, BN, false, 0, false, true, false, false);
            return true;
        }
    }
    if constexpr ((kMask >> kConvSmallCin) & 1u) {
        if (variant == kConvSmallCin) {
            FIERY_CONV_LAUNCH(BM, BN, false, 0, true, false, false, false);
            return true;
        }
    }
#if FIERY_CONV_TUNING
    if constexpr ((kMask >> kConvAligned) & 1u) {            // the probes travel with the scalar-addressed kernel's unit
        if (variant == kConvClock || variant == kConvClockAligned) {
            if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_clk_probe), &clk, sizeof(clk), 0, hipMemcpyHostToDevice, hs) != hipSuccess)
                return false;
            if (variant == kConvClockAligned) FIERY_CONV_LAUNCH(BM, BN, true, 0, false, true, false, false);
            else FIERY_CONV_LAUNCH(BM, BN, true, 0, false, false, false, false);
            return true;
        }
        if (variant == kConvPrio) {
            FIERY_CONV_LAUNCH(BM, BN, false, 1, false, false, false, false);
            return true;
        }
    }
#endif
    return false;
}

}  // namespace
#endif  // FIERY_CONV_KERNEL_TU

}  // namespace fiery
