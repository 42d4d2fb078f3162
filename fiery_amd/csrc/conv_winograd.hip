// Winograd F(2x2, 3x3) form of the 3 x 3 / stride 1 / 'same' convolutions of the BEV stack, fp32 on the gfx950 matrix cores
// (round 5).  Reference layers: fiery/layers/convolutions.py:9-60 (ConvBlock), fiery/layers/temporal.py:36-62 (SpatialGRU's
// gate and candidate convolutions), fiery/models/decoder.py:53-91 (BasicBlock / upsampling stages) - everything `conv2d(k = 3,
// stride = 1, padding = 1)` with 64 couts or more.
//
// WHY.  The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector ALU's multiplier rate, 157.3 TFLOP/s; four rounds of tuning
// left the direct implicit GEMM at 0.65-0.75 of it on these layers, which hold ~85 % of the step's convolution time.  What is
// left is doing fewer multiplies: F(2x2, 3x3) computes a 2 x 2 output block from a 4 x 4 input block with 16 multiplies per
// (cin, cout) pair instead of 36 - 2.25x fewer - at the price of add-only transforms of the input (32 adds per block and
// channel, shared by all couts) and of the output (24 adds per block and cout, shared by all cins).
//   Y = A^T [ sum_c (G g_c G^T) . (B^T d_c B) ] A
// Numerics were gated first (tests/experiments/winograd_numerics.py, profiles/r5_winograd_numerics.txt): the oracle's hot
// path with every such convolution through an fp32 emulation of this form lands 7.9e-6 from the direct fp32 evaluation on
// the worst output (the bar is 1e-4) and at the same distance from the fp64 evaluation (7.00e-4 vs 6.98e-4).
//
// HOW.  16 independent GEMMs, one per transform point p: M_p[tile][cout] = sum_c V_p[tile][c] U_p[c][cout].
//   * workgroup = 4 wavefronts, 32 tiles (= 128 output pixels) x 64 couts; wavefront w owns the transform points 4 w .. 4 w + 3
//     for all 32 tiles and both 32-cout blocks: 8 accumulator blocks = 128 registers, so two workgroups fit a CU and one's
//     transforms / epilogue run under the other's MFMAs.
//   * K advances 16 channels per stage.  Every thread owns one (tile, channel pair) of the stage: it fetches the tile's 4 x 4
//     input block for its two channels straight from the pixel-major activations (16 eight-byte buffer loads, border taps
//     masked to the descriptor's out-of-range zero), transforms it in registers with packed adds and writes the sixteen
//     V_p values to LDS ([p][tile][16 k], 16-byte slots XOR-swizzled so the operand reads are conflict-free).  The loads of
//     stage s + 1 are in flight while stage s is multiplied; LDS is double-buffered, one barrier per stage.
//   * the transformed weights U (host-side: G g G^T evaluated in fp64, rounded once - fiery_conv_pack_weights_winograd) never
//     touch LDS: a lane's operand of four MFMA k-steps is one 16-byte piece of the packed image [p][k / 4][cout][k % 4], and
//     each wavefront requests exactly the pieces of its own transform points into a register ring, several pieces ahead.
//   * MFMA operands are swapped (weights as A, tiles as B), so an accumulator block is lane = tile, registers = couts: four
//     consecutive couts sit in four consecutive registers and go to LDS as 16-byte pieces.
//   * epilogue: per 32-cout block the sixteen M_p blocks meet in LDS ([p][tile][cout], the K loop's buffers reused); a thread
//     takes (tile, four couts), applies A^T . A, then bias / folded BatchNorm / activation / residual or the GRU gate
//     arithmetic of the direct kernel's epilogues, and stores 16 bytes to each of the block's four pixels.
// Executed matrix flops are 16 / 36 of the direct form's; bench.py reports the algorithmic (direct-form) flops of these
// launches and the executed-MFMA fraction side by side, never an "algorithmic TFLOP/s" against the matrix peak.
//
// SPLIT form (round 6; conv_winograd_split.hip compiles this file with FIERY_WINOGRAD_SPLIT = 1).  The 16 GEMMs run on the bf16
// matrix cores (v_mfma_f32_32x32x16_bf16: sixteen times the fp32 instruction's rate) WITHOUT giving up fp32 accuracy: every fp32
// operand is written as the sum of three bf16 terms (x = x1 + x2 + x3 exactly: 3 x 8 significand bits) and the product as the
// six partial products whose weight is not below 2^-24 of it - u1 v3, u3 v1, u2 v2, u1 v2, u2 v1, u1 v1, smallest first - each
// exact in the fp32 accumulator.  Measured against fp64 on this chip (tools/probe/split_bf16_probe.hip, profiles/
// r6_split_bf16_probe.txt): rms error 3.6e-7 of the result's rms at K = 576 where the fp32 instruction leaves 4.3e-7 (the bf16
// instruction adds sixteen products per rounding, the fp32 one two), 1.07e-6 against 1.21e-6 at K = 4608; nine products change
// nothing.  Six 8-pass MFMAs per sixteen channels replace eight 16-pass ones: 192 against 512 matrix-pipe cycles.  The
// transformed weights are split on the host side of the launch (k_pack_winograd_split: U in fp64, rounded to fp32, then split);
// the transformed input stays fp32 in LDS and the wavefront that owns a transform point splits its operand as it reads it
// (36 vector instructions per point and stage, which run under the other wavefront's MFMAs: tools/probe/
// mfma_bf16_valu_probe.hip).  Everything outside the K loop - input transform, epilogues - is the fp32 form's.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"
#ifndef FIERY_WINOGRAD_SPLIT
#define FIERY_WINOGRAD_SPLIT 0
#endif

namespace fiery {
namespace {

#ifndef W_RING
#define W_RING 16                 // register ring of weight pieces (8 or 16: it must divide the 16 pieces of a stage)
#endif
#ifndef W_DEFAULT_WAVES
#define W_DEFAULT_WAVES 4         // wavefronts per workgroup unless FIERY_WINOGRAD_WAVES says otherwise (4 or 8)
#endif
#ifndef W_RING8
#define W_RING8 4                 // the ring of the eight-wavefront form (8 pieces per stage and wavefront; 8 spills at 128 registers)
#endif
#ifndef W_INTERLEAVE
#define W_INTERLEAVE 0            // the next stage's input transform in one piece after the first half of the MFMAs (1: dealt out between the blocks - measured 1-2 % slower, profiles/r5_winograd_variants.txt)
#endif
#ifndef W_TRACE
#define W_TRACE 0                 // tuning builds: per-workgroup timeline (100 MHz wall clock at entry / first barrier / end of the K
#endif                            // loop / end, + the CU's hardware id) into the buffer whose address rides in p.sk_ws
#ifndef W_PIPE
#define W_PIPE 1                  // pin the MFMA order of a transform point's block (A/B switch)
#endif
#ifndef W_SPLIT_RING
#define W_SPLIT_RING 12           // split form: register ring of weight pieces (a stage has 24: 4 points x 2 cout blocks x 3 terms)
#endif
#ifndef W_SPLIT_TRANSFORM_AT
#define W_SPLIT_TRANSFORM_AT 1    // split form: the next stage's input transform follows this point's MFMAs (0 .. 3)
#endif
#ifndef W_EXP
#define W_EXP 0                   // timing experiments (wrong results): 1 weights from one hot piece, 2 no input loads in the loop,
#endif                            // 3 no transform / V stores in the loop, 4 no epilogue
constexpr int WT = 32;            // tiles (2 x 2 output blocks) per workgroup
constexpr int WBN = 64;           // couts per workgroup
constexpr int WKC = 16;           // channels per stage
constexpr int W_V_FLOATS = 16 * WT * WKC;              // one stage of V
constexpr int W_M_PITCH = 36;                          // floats between tiles of the epilogue's exchange block (32 couts + 4)
constexpr int W_SMEM_FLOATS = 16 * WT * W_M_PITCH;     // 73,728 bytes: the exchange block; the two V stages (65,536) lie inside
static_assert(2 * W_V_FLOATS <= W_SMEM_FLOATS, "the V stages must fit the block");

// floats of the transformed, packed weights: [cout tile][p][cin_pad / 4][64][4]
__host__ __device__ inline long long wino_packed_floats(int cout_pad64, int cin_pad) { return 16ll * cout_pad64 * cin_pad; }

// U = G g G^T in fp64, rounded once; packed[(((tn * 16 + p) * Q + k / 4) * 64 + n % 64) * 4 + k % 4], k = padded channel position
__global__ void k_pack_winograd(const float* __restrict__ w, int cout, int cin_total, ChanInverse inv, int cin_units, long long total,
                                float* __restrict__ packed) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int Q = cin_units * 2;
    const int kk = static_cast<int>(i & 3);
    long long r = i >> 2;
    const int nn = static_cast<int>(r % 64);
    r /= 64;
    const int quad = static_cast<int>(r % Q);
    r /= Q;
    const int p = static_cast<int>(r % 16);
    const int tn = static_cast<int>(r / 16);
    const int n = tn * 64 + nn, k = quad * 4 + kk;
    float v = 0.f;
    if (n < cout) {
        const int ci = inv.ci[k];
        if (ci >= 0) {
            const float* g = w + (static_cast<long long>(n) * cin_total + ci) * 9;
            const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
            const int pi = p >> 2, pj = p & 3;
            double s = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) s += G[pi][a] * static_cast<double>(g[a * 3 + b]) * G[pj][b];
            v = static_cast<float>(s);
        }
    }
    packed[i] = v;
}

// Split form: U as above, rounded to fp32, then written as three bf16 terms (round to nearest even; the three add up to the fp32
// value exactly).  Packed as the 16-byte MFMA operands of the kernel's lanes: [cout tile][p][stage][cout block][term][lane][8],
// lane = (cout % 32, hi), element j = channel 16 stage + (j < 4 ? 4 hi + j : 8 + 4 hi + j - 4) - the channels lane (tile, hi) of
// the V operand holds (slots hi and 2 + hi of the tile's row).
__global__ void k_pack_winograd_split(const float* __restrict__ w, int cout, int cin_total, ChanInverse inv, int cin_units, long long total,
                                      unsigned short* __restrict__ packed) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int S = cin_units >> 1;
    const int j = static_cast<int>(i & 7), lane = static_cast<int>((i >> 3) & 63);
    long long r = i >> 9;
    const int term = static_cast<int>(r % 3);
    r /= 3;
    const int nb = static_cast<int>(r & 1);
    r >>= 1;
    const int st = static_cast<int>(r % S);
    r /= S;
    const int p = static_cast<int>(r % 16);
    const int tn = static_cast<int>(r / 16);
    const int hi = lane >> 5, n = tn * 64 + nb * 32 + (lane & 31);
    const int k = 16 * st + (j < 4 ? 4 * hi + j : 8 + 4 * hi + j - 4);
    float v = 0.f;
    if (n < cout) {
        const int ci = inv.ci[k];
        if (ci >= 0) {
            const float* g = w + (static_cast<long long>(n) * cin_total + ci) * 9;
            const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
            const int pi = p >> 2, pj = p & 3;
            double sum = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) sum += G[pi][a] * static_cast<double>(g[a * 3 + b]) * G[pj][b];
            v = static_cast<float>(sum);
        }
    }
    const float t1 = bf16_round(v), r1 = v - t1, t2 = bf16_round(r1), r2 = r1 - t2;
    packed[i] = bf16_bits(term == 0 ? t1 : term == 1 ? t2 : r2);
}

// KIND: which epilogue the kernel carries - with all of them behind run-time switches the code after the K loop was 18,000
// instructions with scalar registers spilled to vector lanes, and an epilogue's vector instructions are served one per MFMA of the
// CU's other workgroup (~64 cycles each): 0 plain, act none / ReLU, no residual, no bias; 1 plain, anything; 2 GRU gates; 3 GRU
// output (0-3: every tensor dense over its images - the lean path); 4 decoder heads; -1 everything, general addressing (fallback)
// NW: wavefronts per workgroup.  4: a wavefront owns four transform points (128 accumulator registers, two wavefronts per SIMD).
// 8 (round 5, second form): two points per wavefront - 64 accumulator registers, the kernel held to 128 registers, FOUR
// wavefronts per SIMD: twice as many MFMA streams to fill the gaps a stream leaves at its barriers and operand waits; a thread
// then transforms one channel of a block (4-byte pieces) instead of two, and the first four wavefronts run the epilogue's rows.
// (SPLIT: a template parameter so that the two forms have different kernel names in profiles)
template <int KIND, int NW, bool SPLIT = (FIERY_WINOGRAD_SPLIT != 0)>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void k_conv_winograd(ConvP p) {
    static_assert(NW == 4 || NW == 8, "four or eight wavefronts");
    constexpr int PP = 16 / NW;                        // transform points per wavefront
    constexpr bool LEAN = KIND >= 0 && KIND <= 4;
    // (SPLIT: bf16 matrix cores, three-term operands - see the head of the file)
    static_assert(!SPLIT || NW == 4, "the split form: four wavefronts");
    __shared__ __attribute__((aligned(16))) float smem[W_SMEM_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int tile_n = blockIdx.y;
    const int H = p.Hout, W = p.Wout;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int tiles_img = TH * TW;
    const int n_tiles = p.n_img * tiles_img;
    // XCD-aware order of the workgroups' tile blocks (workgroup b runs on XCD b % 8: a contiguous run of blocks per XCD)
    int blk;
    {
        const int nblk = static_cast<int>(gridDim.x), bid = static_cast<int>(blockIdx.x);
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile0 = blk * WT;
#if W_TRACE
    unsigned long long* trace = nullptr;
    if (p.sk_ws && tid == 0) {
        trace = reinterpret_cast<unsigned long long*>(p.sk_ws) + 8ull * (blockIdx.x + static_cast<unsigned long long>(blockIdx.y) * gridDim.x);
        trace[0] = wall_clock64();
        trace[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        trace[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    }
#endif

    // ---- this thread's input block: tile tt, channels 2 cp, 2 cp + 1 of the stage ------------------------------------------
    // (NW = 4: tile tt, channel PAIR cp of the stage, 8-byte pieces; NW = 8: tile tt, ONE channel cp, 4-byte pieces)
    using DT = std::conditional_t<NW == 4, v2f, float>;
    constexpr int CPT = NW == 4 ? 2 : 1;               // channels per thread
    const int tt = NW == 4 ? tid >> 3 : tid >> 4, cp = NW == 4 ? tid & 7 : tid & 15;
    int e_o, e_y, e_x;              // the tile's image and top-left output pixel (kept for the epilogue)
    bool e_live;
    int voff0, voff1;               // byte offsets of the block's top-left tap (y0 - 1 + 1 lead row ...) in the two sources
    unsigned tapmask = 0;           // bit 4 i + j: tap (i, j) lies inside the image
    {
        const int T = tile0 + tt;
        const bool live = T < n_tiles;
        const int Tq = live ? T : 0;
        const int o = fast_div(Tq, p.wmg_img, p.wsh_img), rem = Tq - o * tiles_img;
        const int ty = fast_div(rem, p.wmg_tw, p.wsh_tw), tx = rem - ty * TW;
        const int b = fast_div(o, p.mg_t, p.sh_t), tl = o - b * p.Tout;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        e_live = live;                                     // (the epilogue's thread-to-tile mapping is this one)
        e_o = o;
        e_y = 2 * ty;
        e_x = 2 * tx;
        if (live) {                                        // valid rows x valid columns (ranges, no loop over the sixteen taps)
            unsigned colm = 0, rowm = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                colm |= static_cast<unsigned>(x0 + j) < static_cast<unsigned>(W) ? 1u << j : 0u;
                rowm |= static_cast<unsigned>(y0 + j) < static_cast<unsigned>(H) ? 1u << j : 0u;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) tapmask |= (rowm >> i) & 1u ? colm << (4 * i) : 0u;
        }
        // (the descriptors start one row and one pixel before the tensors: the top-left tap's offset is never negative)
        const int pos = (y0 + 1) * W + (x0 + 1);
        voff0 = 4 * (b * static_cast<int>(p.src[0].bstride) + (tl + p.tinadd) * static_cast<int>(p.src[0].tstride) + pos * p.src[0].ld + CPT * cp);
        voff1 = 4 * (b * static_cast<int>(p.src[1].bstride) + (tl + p.tinadd) * static_cast<int>(p.src[1].tstride) + pos * p.src[1].ld + CPT * cp);
    }
    const int lead0 = (W + 1) * p.src[0].ld, lead1 = (W + 1) * p.src[1].ld;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src[0].ptr - lead0), 0, 4 * lead0 + p.src[0].ext_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.src[1].ptr ? p.src[1].ptr - lead1 : p.src[0].ptr), 0, p.src[1].ptr ? 4 * lead1 + p.src[1].ext_bytes : 0, 0x00020000);
    const int stages = p.cin_units >> 1, stages0 = p.src[0].units >> 1;
    const int rowb0 = W * p.src[0].ld * 4, pixb0 = p.src[0].ld * 4, rowb1 = W * p.src[1].ld * 4, pixb1 = p.src[1].ld * 4;

    DT d[16];                       // the 4 x 4 block (one or two channels per element), transformed in place
    // (a wavefront whose eight tiles all lie inside the image - almost all do - needs no per-tap select: 16 vector instructions
    // per stage less, and vector instructions are what the K loop pays for)
    const bool interior = __ballot(tapmask != 0xFFFFu) == 0ull;
    auto request = [&](int s) {
        const bool second = s >= stages0;
        const int ch = 64 * (s - (second ? stages0 : 0));                  // bytes: 16 channels per stage
        const int rowb = second ? rowb1 : rowb0, pixb = second ? pixb1 : pixb0;
        const int vo = second ? voff1 : voff0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = (interior || ((tapmask >> (4 * i + j)) & 1u)) ? vo : static_cast<int>(0x80000000u);     // outside: reads as zero
                if constexpr (NW == 4) {
                    const auto raw = second ? __builtin_amdgcn_raw_buffer_load_b64(rs1, v, ch + i * rowb + j * pixb, 0)
                                            : __builtin_amdgcn_raw_buffer_load_b64(rs0, v, ch + i * rowb + j * pixb, 0);
                    __builtin_memcpy(&d[4 * i + j], &raw, 8);
                } else {
                    const unsigned raw = second ? __builtin_amdgcn_raw_buffer_load_b32(rs1, v, ch + i * rowb + j * pixb, 0)
                                                : __builtin_amdgcn_raw_buffer_load_b32(rs0, v, ch + i * rowb + j * pixb, 0);
                    __builtin_memcpy(&d[4 * i + j], &raw, 4);
                }
            }
    };
    // B^T d B in place (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]): columns, then rows - 32 packed adds, in four parts
    auto transform_part = [&](int part) {
        if (part < 2) {
#pragma unroll
            for (int j = 2 * part; j < 2 * part + 2; ++j) {
                const DT d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
                d[j] = d0 - d2;
                d[4 + j] = d1 + d2;
                d[8 + j] = d2 - d1;
                d[12 + j] = d1 - d3;
            }
        } else {
#pragma unroll
            for (int i = 2 * (part - 2); i < 2 * (part - 2) + 2; ++i) {
                const DT t0 = d[4 * i], t1 = d[4 * i + 1], t2 = d[4 * i + 2], t3 = d[4 * i + 3];
                d[4 * i] = t0 - t2;
                d[4 * i + 1] = t1 + t2;
                d[4 * i + 2] = t2 - t1;
                d[4 * i + 3] = t1 - t3;
            }
        }
    };
    auto transform = [&]() {
#pragma unroll
        for (int part = 0; part < 4; ++part) transform_part(part);
    };
    // V stage in LDS: [p][tile][16 k]; 16-byte slot q of a tile's row sits at q ^ ((tile >> 2) & 3)
    const int v_st = NW == 4 ? (tt * WKC + 4 * ((cp >> 1) ^ ((tt >> 2) & 3)) + 2 * (cp & 1))
                             : (tt * WKC + 4 * ((cp >> 2) ^ ((tt >> 2) & 3)) + (cp & 3));                // + p * WT * WKC + buf * W_V_FLOATS
    auto store_v_rows = [&](int buf, int i0, int i1) {                   // rows [i0, i1) of the transformed block: points 4 i .. 4 i + 3
#pragma unroll
        for (int pp = 4 * i0; pp < 4 * i1; ++pp) *reinterpret_cast<DT*>(&smem[buf * W_V_FLOATS + pp * (WT * WKC) + v_st]) = d[pp];
    };
    auto store_v = [&](int buf) { store_v_rows(buf, 0, 4); };
    // operand reads: lane (m, hi) of k-group q reads slot 2 q + hi of tile m's row
    int v_rd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) v_rd[q] = m * WKC + 4 * ((2 * q + hi) ^ ((m >> 2) & 3));          // + p * WT * WKC + buf * W_V_FLOATS

    // ---- transformed weights: this wavefront's pieces, straight into a register ring -----------------------------------------
    // piece index within a stage: ((q * 4 + pl) * 2 + nb), pl = local transform point; 16 pieces per stage
    // (split form: a stage has 24 pieces - piece 6 pl + 3 nb + term of [p][stage][cout block][term][lane], 1 KiB each)
    const int Q = p.cin_units * 2;                                      // 16-byte k-quads of the whole K
    const __amdgpu_buffer_rsrc_t wrs = SPLIT
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.w)) + static_cast<long long>(tile_n) * 16 * stages * 6144, 0,
                                            16 * stages * 6144, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + static_cast<long long>(tile_n) * 16 * Q * 256, 0, 16 * Q * 1024, 0x00020000);
    const int w_vo = SPLIT ? lane * 16 : (hi * 64 + m) * 16;
    // (the ring is deep: vmcnt counts loads in issue order, so a weight piece requested behind the next stage's 16 input loads
    // cannot be waited for without waiting for those too - pieces are requested almost a stage before they are multiplied, by
    // which time the input loads in front of them have long landed)
    constexpr int PIECES = SPLIT ? 6 * PP : 4 * PP;                    // per stage: two k-groups x PP points x two cout halves
    // (split form: a point's six pieces are requested into the slots the point before last just left)
    constexpr int RING = SPLIT ? W_SPLIT_RING : NW == 4 ? W_RING : W_RING8, AHEAD = SPLIT ? RING : RING - 2;
    static_assert(PIECES % RING == 0, "the ring must divide a stage's pieces");
    float4 wr[RING];
    auto to_f4 = [](auto raw) {
        float4 f;
        __builtin_memcpy(&f, &raw, 16);
        return f;
    };
    bool w_live = true;                                                 // (W_EXP == 6: no weight loads after the prologue)
    auto w_request = [&](int slot, int s, int piece) {                  // piece of stage s (past the end: any piece of the last stage)
        if (W_EXP == 6 && !w_live) return;
        const int ss = s < stages ? s : stages - 1;
        if constexpr (SPLIT) {
            const int soff = W_EXP == 1 ? 0 : (((PP * wv + piece / 6) * stages + ss) * 6 + piece % 6) * 1024;
            wr[slot] = to_f4(__builtin_amdgcn_raw_buffer_load_b128(wrs, w_vo, soff, 0));
            return;
        }
        const int q = piece / (2 * PP), pl = (piece >> 1) % PP, nb = piece & 1;
        const int soff = W_EXP == 1 ? 0 : (((PP * wv + pl) * Q + 4 * ss + 2 * q) * 64 + nb * 32) * 16;
        wr[slot] = to_f4(__builtin_amdgcn_raw_buffer_load_b128(wrs, w_vo, soff, 0));
    };

    // (no zero fill: the first MFMA of every accumulator block takes the constant 0 as its addend - 128 vector moves less in the
    // prologue, which runs at one instruction per MFMA of the CU's other workgroup)
    v16f acc[PP][2];
    // ---- prologue: stage 0 into LDS, stage 1's block and the first weight pieces in flight ----------------------------------
    request(0);
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) w_request(i % RING, i < PIECES ? 0 : 1, i < PIECES ? i : i - PIECES);
    transform();
    store_v(0);
    if (stages > 1) request(1);
    __syncthreads();
#if W_TRACE
    if (trace) trace[1] = wall_clock64();
#endif

    auto comp = [](const float4& f, int j) { return j == 0 ? f.x : j == 1 ? f.y : j == 2 ? f.z : f.w; };
    v16f zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    auto stage = [&](int s, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int buf = s & 1;
        const float* vb = smem + buf * W_V_FLOATS + (PP * wv) * (WT * WKC);
        if constexpr (SPLIT) {
            // a stage is ONE sixteen-channel MFMA step per (point, cout block, product): the lane's eight channels of its tile
            // are slots hi and 2 + hi of the row (two 16-byte reads, made a point ahead), split here into the three terms
            float4 n_lo = *reinterpret_cast<const float4*>(vb + v_rd[0]), n_hi = *reinterpret_cast<const float4*>(vb + v_rd[1]);
#pragma unroll
            for (int pl = 0; pl < PP; ++pl) {
                const float4 lo = n_lo, up = n_hi;
                if (pl + 1 < PP) {
                    n_lo = *reinterpret_cast<const float4*>(vb + (pl + 1) * (WT * WKC) + v_rd[0]);
                    n_hi = *reinterpret_cast<const float4*>(vb + (pl + 1) * (WT * WKC) + v_rd[1]);
                }
                const float x[8] = {lo.x, lo.y, lo.z, lo.w, up.x, up.y, up.z, up.w};
                bf16x8 v1, v2, v3;
                split_bf16x8(x, v1, v2, v3);
                bf16x8 u[2][3];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int t = 0; t < 3; ++t) u[nb][t] = bits_bf16x8(wr[(6 * pl + 3 * nb + t) % RING]);
#if W_PIPE
                __builtin_amdgcn_sched_barrier(0);
#endif
                // smallest products first: u1 v3, u3 v1, u2 v2, u1 v2, u2 v1, u1 v1; the two cout blocks alternate
                constexpr int TU[6] = {0, 2, 1, 0, 1, 0}, TV[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const bf16x8 vv = TV[k] == 0 ? v1 : TV[k] == 1 ? v2 : v3;
                    const bool fresh = FIRST && k == 0;
                    acc[pl][0] = mfma_bf16_32x32x16(u[0][TU[k]], vv, fresh ? zero16 : acc[pl][0]);
                    acc[pl][1] = mfma_bf16_32x32x16(u[1][TU[k]], vv, fresh ? zero16 : acc[pl][1]);
#if W_PIPE
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int nxt = 6 * pl + e + AHEAD;
                    if (nxt < PIECES) w_request(nxt % RING, s, nxt);
                    else w_request(nxt % RING, s + 1, nxt - PIECES);
                }
                if (s + 1 < stages && pl == W_SPLIT_TRANSFORM_AT) {
                    if (W_EXP != 3) transform();
                    if (W_EXP != 3) store_v(buf ^ 1);
                    if (s + 2 < stages && W_EXP != 2) request(s + 2);
                }
            }
            __syncthreads();
            return;
        }
        // (the block operand of point pl + 1 is read from LDS while the MFMAs of point pl run; with W_PIPE the MFMA order is pinned
        // as written - two accumulator blocks alternating - instead of the four-long dependent chains the scheduler prefers)
        float4 a_nxt = *reinterpret_cast<const float4*>(vb + v_rd[0]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int pl = 0; pl < PP; ++pl) {
                const float4 a4 = a_nxt;
                if (q * PP + pl < 2 * PP - 1) a_nxt = *reinterpret_cast<const float4*>(vb + ((pl + 1) % PP) * (WT * WKC) + v_rd[pl == PP - 1 ? q + 1 : q]);
                const int piece = (q * PP + pl) * 2;
                const float4 b0 = wr[piece % RING], b1 = wr[(piece + 1) % RING];
#if W_PIPE
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool fresh = FIRST && q == 0 && j == 0;
                    acc[pl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(b0, j), comp(a4, j), fresh ? zero16 : acc[pl][0], 0, 0, 0);
                    acc[pl][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(b1, j), comp(a4, j), fresh ? zero16 : acc[pl][1], 0, 0, 0);
#if W_PIPE
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                // the two slots this point leaves take the pieces AHEAD further on (wrapping into the next stage)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int nxt = piece + e + AHEAD;
                    if (nxt < PIECES) w_request(nxt % RING, s, nxt);
                    else w_request(nxt % RING, s + 1, nxt - PIECES);
                }
                // the next stage's block is transformed and stored in the first half of this one (its loads were requested half a
                // stage ago), and the block after it requested as soon as the registers are free
                if (s + 1 < stages) {
#if W_INTERLEAVE
                    static_assert(PP == 4, "the interleaved transform is dealt over four points");
                    if (q == 0) {
                        if (W_EXP != 3) transform_part(pl);          // columns 0-1, 2-3, rows 0-1, 2-3
                        if (pl == 2 && W_EXP != 3) store_v_rows(buf ^ 1, 0, 2);
                        if (pl == 3) {
                            if (W_EXP != 3) store_v_rows(buf ^ 1, 2, 4);
                            if (s + 2 < stages && W_EXP != 2) request(s + 2);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#else
                    if (q == 0 && pl == PP - 1) {
                        if (W_EXP != 3) transform();
                        if (W_EXP != 3) store_v(buf ^ 1);
                        if (s + 2 < stages && W_EXP != 2) request(s + 2);
                    }
#endif
                }
            }
        }
        __syncthreads();
    };
    if (W_EXP == 6) w_live = false;
    stage(0, std::true_type{});
    for (int s = 1; s < stages; ++s) stage(s, std::false_type{});
#if W_TRACE
    if (trace) trace[2] = wall_clock64();
#endif

    // ---- epilogue: per 32-cout block, M_p -> LDS [p][tile][cout], then A^T M A and the direct kernel's epilogue arithmetic -------
    // this thread's tile and four couts of the block: 256 (tile, cout quad) tasks - all threads of the four-wavefront form (the
    // transform's own mapping), the first four wavefronts of the eight-wavefront form (the others are done after the exchange)
    const int et = NW == 4 ? tt : (tid >> 3) & 31, cq = tid & 7;
    const bool e_worker = NW == 4 || wv < 4;
    if constexpr (NW == 8) {
        const int T = tile0 + et;
        e_live = T < n_tiles && e_worker;
        const int Tq = T < n_tiles ? T : 0;
        e_o = fast_div(Tq, p.wmg_img, p.wsh_img);
        const int rem = Tq - e_o * tiles_img;
        const int ty = fast_div(rem, p.wmg_tw, p.wsh_tw);
        e_y = 2 * ty;
        e_x = 2 * (rem - ty * TW);
    }
    const int half = p.cout_pad >> 1;
    auto activate = [](float v, int act) {
        if (act == FIERY_ACT_RELU) return fmaxf(v, 0.f);
        if (act == FIERY_ACT_SIGMOID) return sigmoidf(v);
        if (act == FIERY_ACT_SWISH) return v * sigmoidf(v);
        return v;
    };
    // FIERY_EPI_HEADS (models/decoder.py:30-51: Conv3x3 -> BN -> ReLU -> Conv1x1 (+ bias) [-> Sigmoid]): a workgroup's 64 couts are
    // one head's hidden channels, which never leave the chip - a thread keeps, for its block's four pixels, the partial dot
    // products of its hidden channels with the rows of the final 1x1 that read this group; the eight threads of a tile add them
    // up at the end and one of them stores the rows' values to their pixel-contiguous planes
    const bool heads = KIND == 4 || (KIND < 0 && p.epi == FIERY_EPI_HEADS);
    const int epi = KIND == 0 || KIND == 1 ? FIERY_EPI_PLAIN : KIND == 2 ? FIERY_EPI_GRU_GATES : KIND == 3 ? FIERY_EPI_GRU_OUT : p.epi;
    float hp[FIERY_MAX_HEAD_OUTPUTS][4];
#pragma unroll
    for (int o = 0; o < FIERY_MAX_HEAD_OUTPUTS; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) hp[o][q] = 0.f;
    auto epilogue_block = [&](auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        // (the K loop's last barrier has passed: the V stages are free)
#pragma unroll
        for (int pl = 0; pl < PP; ++pl)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v16f& a = acc[pl][nb];
                *reinterpret_cast<float4*>(&smem[((PP * wv + pl) * WT + m) * W_M_PITCH + 8 * g + 4 * hi]) =
                    make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
            }
        __syncthreads();
        if (!e_worker) return;                                      // (eight-wavefront form: the first four take the rows)
        const int co = tile_n * WBN + nb * 32 + 4 * cq;
        const bool upper = epi == FIERY_EPI_GRU_GATES && co >= half;
        const int c_x = upper ? co - half : co;
        if constexpr (LEAN) {
            // ---- the lean path (every tensor dense over its images, 16-byte rows): the epilogue's vector instructions are
            // served one per MFMA of the CU's other workgroup (~64 cycles each), so their NUMBER is what
            // the launch pays for - packed adds for the transform (48 per 16 outputs), one buffer offset per tensor with the
            // block's four pixels as scalar offsets, no per-pixel address arithmetic.
            const bool ch_ok = e_live && (KIND == 4 || c_x < p.cout_store);
            v2f mlo[16], mhi[16];
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                const float4 f = *reinterpret_cast<const float4*>(&smem[(pp * WT + et) * W_M_PITCH + 4 * cq]);
                mlo[pp] = v2f{f.x, f.y};
                mhi[pp] = v2f{f.z, f.w};
            }
            // A^T M A with A^T = [1 1 1 0; 0 1 -1 -1]: s = m1 + m2, d = m1 - m2, (m0 + s, d - m3) - columns, then rows
            v2f ylo[4], yhi[4];
            auto out_transform = [](const v2f (&mm)[16], v2f (&y)[4]) {
                v2f t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f sm = mm[4 + j] + mm[8 + j], df = mm[4 + j] - mm[8 + j];
                    t0[j] = mm[j] + sm;
                    t1[j] = df - mm[12 + j];
                }
                { const v2f sm = t0[1] + t0[2], df = t0[1] - t0[2]; y[0] = t0[0] + sm; y[1] = df - t0[3]; }
                { const v2f sm = t1[1] + t1[2], df = t1[1] - t1[2]; y[2] = t1[0] + sm; y[3] = df - t1[3]; }
            };
            out_transform(mlo, ylo);
            out_transform(mhi, yhi);
            const float4 sc = *reinterpret_cast<const float4*>(p.scale + co), sh = *reinterpret_cast<const float4*>(p.shift + co);
            const v2f sclo = v2f{sc.x, sc.y}, schi = v2f{sc.z, sc.w}, shlo = v2f{sh.x, sh.y}, shhi = v2f{sh.z, sh.w};
            // this thread's byte offset of pixel (e_y, e_x) of image e_o in a tensor (image stride, row length ld), channel c - 31 bits
            // (conv_run checked the spans); the block's other pixels are scalar offsets away; pixels outside an odd-sized image
            // (and tiles past the end) are pointed out of range
            const int pix00 = e_y * W + e_x;
            const bool v01 = e_x + 1 < W, v10 = e_y + 1 < H;
            const bool whole = __ballot(!(v01 && v10)) == 0ull;          // (even image sizes: every block of the wavefront is whole)
            auto rsrc_of = [&](const TensP& t) {
                const int bytes = t.ptr ? ((p.n_img - 1) * static_cast<int>(t.istride) + H * W * t.ld) * 4 : 0;
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.ptr), 0, bytes, 0x00020000);
            };
            auto off_of = [&](const TensP& t, int c) {
                return ch_ok ? (e_o * static_cast<int>(t.istride) + pix00 * t.ld + c) * 4 : static_cast<int>(0x80000000u);
            };
            auto ld4 = [&](__amdgpu_buffer_rsrc_t r, int vo, int q, int ld_) {
                const bool ok = whole || q == 0 || (q == 1 ? v01 : q == 2 ? v10 : (v01 && v10));
                return to_f4(__builtin_amdgcn_raw_buffer_load_b128(r, ok ? vo : static_cast<int>(0x80000000u), (q >> 1) * W * ld_ * 4 + (q & 1) * ld_ * 4, 0));
            };
            auto st4 = [&](__amdgpu_buffer_rsrc_t r, int vo, int q, int ld_, const float4& f) {
                const bool ok = whole || q == 0 || (q == 1 ? v01 : q == 2 ? v10 : (v01 && v10));
                decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) raw;
                __builtin_memcpy(&raw, &f, 16);
                __builtin_amdgcn_raw_buffer_store_b128(raw, r, ok ? vo : static_cast<int>(0x80000000u), (q >> 1) * W * ld_ * 4 + (q & 1) * ld_ * 4, 0);
                store_data_settle();                                // (the next pixel's packed arithmetic lands in the same registers)
            };
            const __amdgpu_buffer_rsrc_t r_out = rsrc_of(p.out);
            const int o_out = off_of(p.out, co);
            // per-image bias rows (the first SpatialGRU's folded constant input: plain AND gate / output epilogues carry it), with
            // the nine border classes of a zero-padded 3 x 3 when asked for: added in front of the folded BatchNorm
            if constexpr (KIND != 0) {
                if (p.img_bias) {
                    const float* bias_base = p.img_bias + (p.bias_border ? 9ll : 1ll) * e_o * p.cout_pad + co;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int cls = 0;
                        if (p.bias_border) {
                            const int y = e_y + (q >> 1), x = e_x + (q & 1);
                            cls = (y == 0 ? 0 : y >= H - 1 ? 2 : 1) * 3 + (x == 0 ? 0 : x >= W - 1 ? 2 : 1);
                        }
                        const float4 bz = *reinterpret_cast<const float4*>(bias_base + cls * p.cout_pad);
                        ylo[q] = ylo[q] + v2f{bz.x, bz.y};
                        yhi[q] = yhi[q] + v2f{bz.z, bz.w};
                    }
                }
            }
            // BatchNorm + (ReLU | nothing): max with 0 or with -inf - one instruction either way, no branch
            const float floor_ = p.act == FIERY_ACT_RELU ? 0.f : -__builtin_inff();
            if constexpr (KIND == 4) {
                // decoder heads: hidden = act(BN(conv)) stays in registers; partial dot products with the rows of the final 1x1
                // that read this 64-channel group
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f lo = pk_fma(ylo[q], sclo, shlo), hi2 = pk_fma(yhi[q], schi, shhi);
                    const float vx = fmaxf(pk_lo(lo), floor_), vy = fmaxf(pk_hi(lo), floor_), vz = fmaxf(pk_lo(hi2), floor_), vw = fmaxf(pk_hi(hi2), floor_);
#pragma unroll
                    for (int o = 0; o < FIERY_MAX_HEAD_OUTPUTS; ++o)
                        if (o < p.heads.n_out && p.heads.group[o] == tile_n) {
                            const float4 w4 = *reinterpret_cast<const float4*>(p.heads.w + o * 64 + nb * 32 + 4 * cq);
                            hp[o][q] = fmaf(vw, w4.w, fmaf(vz, w4.z, fmaf(vy, w4.y, fmaf(vx, w4.x, hp[o][q]))));
                        }
                }
                if (!ch_ok) {
#pragma unroll
                    for (int o = 0; o < FIERY_MAX_HEAD_OUTPUTS; ++o)
#pragma unroll
                        for (int q = 0; q < 4; ++q) hp[o][q] = 0.f;
                }
            } else if constexpr (KIND == 0) {
                // (all four pixels are finished before the first store: see store_data_settle)
                float4 fin[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f lo = pk_fma(ylo[q], sclo, shlo), hi2 = pk_fma(yhi[q], schi, shhi);
                    fin[q] = make_float4(fmaxf(pk_lo(lo), floor_), fmaxf(pk_hi(lo), floor_), fmaxf(pk_lo(hi2), floor_), fmaxf(pk_hi(hi2), floor_));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) st4(r_out, o_out, q, p.out.ld, fin[q]);
            } else if constexpr (KIND == 1) {
                const __amdgpu_buffer_rsrc_t r_res = rsrc_of(p.res);
                const int o_res = off_of(p.res, co);
                float4 r4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r4[q] = p.res.ptr ? ld4(r_res, o_res, q, p.res.ld) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v2f lo = ylo[q], hi2 = yhi[q];
                    lo = pk_fma(lo, sclo, shlo);
                    hi2 = pk_fma(hi2, schi, shhi);
                    const v2f rlo = v2f{r4[q].x, r4[q].y}, rhi = v2f{r4[q].z, r4[q].w};
                    if (p.res_pre) { lo = lo + rlo;  hi2 = hi2 + rhi; }
                    float4 v = make_float4(fmaxf(pk_lo(lo), floor_), fmaxf(pk_hi(lo), floor_), fmaxf(pk_lo(hi2), floor_), fmaxf(pk_hi(hi2), floor_));
                    if (!p.res_pre && p.res.ptr) { v.x += r4[q].x;  v.y += r4[q].y;  v.z += r4[q].z;  v.w += r4[q].w; }
                    r4[q] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) st4(r_out, o_out, q, p.out.ld, r4[q]);
            } else if constexpr (KIND == 2) {
                const __amdgpu_buffer_rsrc_t r_h = rsrc_of(p.aux0), r_o2 = rsrc_of(p.out2);
                const int o_h = off_of(p.aux0, c_x), o_o2 = off_of(p.out2, c_x);
                float4 h4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) h4[q] = upper ? ld4(r_h, o_h, q, p.aux0.ld) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f lo = pk_fma(ylo[q], sclo, shlo), hi2 = pk_fma(yhi[q], schi, shhi);
                    float4 g = make_float4(sigmoid_gate(pk_lo(lo)), sigmoid_gate(pk_hi(lo)), sigmoid_gate(pk_lo(hi2)), sigmoid_gate(pk_hi(hi2)));
                    if (upper) {                                                                                                      // (1 - reset) * state
                        g.x = (1.0f - g.x) * h4[q].x;  g.y = (1.0f - g.y) * h4[q].y;  g.z = (1.0f - g.z) * h4[q].z;  g.w = (1.0f - g.w) * h4[q].w;
                    }
                    h4[q] = g;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (upper) st4(r_o2, o_o2, q, p.out2.ld, h4[q]);
                    else st4(r_out, o_out, q, p.out.ld, h4[q]);                                                                       // update gate
                }
            } else {                                                                                                                  // FIERY_EPI_GRU_OUT
                const __amdgpu_buffer_rsrc_t r_u = rsrc_of(p.aux0), r_h = rsrc_of(p.aux1), r_o2 = rsrc_of(p.out2);
                const int o_u = off_of(p.aux0, co), o_h = off_of(p.aux1, co), o_o2 = off_of(p.out2, co);
                float4 u4[4], h4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u4[q] = ld4(r_u, o_u, q, p.aux0.ld);
                    h4[q] = ld4(r_h, o_h, q, p.aux1.ld);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f lo = pk_fma(ylo[q], sclo, shlo), hi2 = pk_fma(yhi[q], schi, shhi);
                    const float4 v = make_float4(pk_lo(lo), pk_hi(lo), pk_lo(hi2), pk_hi(hi2));
                    const float4 u = u4[q], h = h4[q];
                    float4 hn;
                    { const float a1 = (1.0f - u.x) * h.x, b1 = u.x * fmaxf(v.x, 0.f); hn.x = a1 + b1; }
                    { const float a1 = (1.0f - u.y) * h.y, b1 = u.y * fmaxf(v.y, 0.f); hn.y = a1 + b1; }
                    { const float a1 = (1.0f - u.z) * h.z, b1 = u.z * fmaxf(v.z, 0.f); hn.z = a1 + b1; }
                    { const float a1 = (1.0f - u.w) * h.w, b1 = u.w * fmaxf(v.w, 0.f); hn.w = a1 + b1; }
                    h4[q] = hn;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    st4(r_out, o_out, q, p.out.ld, h4[q]);
                    if (p.out2.ptr) st4(r_o2, o_o2, q, p.out2.ld, h4[q]);
                }
            }
            return;
        }
        if constexpr (!LEAN) {
        if (e_live && (heads || c_x < p.cout_store)) {
            float4 mm[16];
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) mm[pp] = *reinterpret_cast<const float4*>(&smem[(pp * WT + et) * W_M_PITCH + 4 * cq]);
            // A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]
            float4 t[2][4];
            auto add3 = [](const float4& a, const float4& b, const float4& c) { return make_float4(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w); };
            auto sub3 = [](const float4& a, const float4& b, const float4& c) { return make_float4(a.x - b.x - c.x, a.y - b.y - c.y, a.z - b.z - c.z, a.w - b.w - c.w); };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = add3(mm[j], mm[4 + j], mm[8 + j]);
                t[1][j] = sub3(mm[4 + j], mm[8 + j], mm[12 + j]);
            }
            const float4 sc = *reinterpret_cast<const float4*>(p.scale + co), sh = *reinterpret_cast<const float4*>(p.shift + co);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int y = e_y + a, x = e_x + b;
                    if (y >= H || x >= W) continue;                     // (odd image sizes: the last row / column of blocks is half outside)
                    float4 v = b == 0 ? add3(t[a][0], t[a][1], t[a][2]) : sub3(t[a][1], t[a][2], t[a][3]);
                    const long long pix = static_cast<long long>(y) * W + x;
                    if (p.img_bias) {
                        long long brow = e_o;
                        if (p.bias_border) brow = brow * 9 + (y == 0 ? 0 : y == H - 1 ? 2 : 1) * 3 + (x == 0 ? 0 : x == W - 1 ? 2 : 1);
                        const float4 bz = *reinterpret_cast<const float4*>(p.img_bias + brow * p.cout_pad + co);
                        v.x += bz.x;  v.y += bz.y;  v.z += bz.z;  v.w += bz.w;
                    }
                    v.x = fmaf(v.x, sc.x, sh.x);  v.y = fmaf(v.y, sc.y, sh.y);  v.z = fmaf(v.z, sc.z, sh.z);  v.w = fmaf(v.w, sc.w, sh.w);
                    if (heads) {
                        v.x = activate(v.x, p.act);  v.y = activate(v.y, p.act);  v.z = activate(v.z, p.act);  v.w = activate(v.w, p.act);
#pragma unroll
                        for (int o = 0; o < FIERY_MAX_HEAD_OUTPUTS; ++o)
                            if (o < p.heads.n_out && p.heads.group[o] == tile_n) {
                                const float4 w4 = *reinterpret_cast<const float4*>(p.heads.w + o * 64 + nb * 32 + 4 * cq);
                                hp[o][2 * a + b] = fmaf(v.w, w4.w, fmaf(v.z, w4.z, fmaf(v.y, w4.y, fmaf(v.x, w4.x, hp[o][2 * a + b]))));
                            }
                    } else if (epi == FIERY_EPI_PLAIN) {
                        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.res.ptr) r = *reinterpret_cast<const float4*>(p.res.ptr + e_o * p.res.istride + pix * p.res.ld + co);
                        if (p.res_pre) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                        v.x = activate(v.x, p.act);  v.y = activate(v.y, p.act);  v.z = activate(v.z, p.act);  v.w = activate(v.w, p.act);
                        if (!p.res_pre) { v.x += r.x;  v.y += r.y;  v.z += r.z;  v.w += r.w; }
                        *reinterpret_cast<float4*>(p.out.ptr + e_o * p.out.istride + pix * p.out.ld + co) = v;
                        store_data_settle();
                    } else if (epi == FIERY_EPI_GRU_GATES) {
                        float4 g = make_float4(sigmoid_gate(v.x), sigmoid_gate(v.y), sigmoid_gate(v.z), sigmoid_gate(v.w));
                        if (!upper) {
                            *reinterpret_cast<float4*>(p.out.ptr + e_o * p.out.istride + pix * p.out.ld + co) = g;                   // update gate
                            store_data_settle();
                        } else {                                                                                                      // (1 - reset) * state
                            const float4 h = *reinterpret_cast<const float4*>(p.aux0.ptr + e_o * p.aux0.istride + pix * p.aux0.ld + c_x);
                            g.x = (1.0f - g.x) * h.x;  g.y = (1.0f - g.y) * h.y;  g.z = (1.0f - g.z) * h.z;  g.w = (1.0f - g.w) * h.w;
                            *reinterpret_cast<float4*>(p.out2.ptr + e_o * p.out2.istride + pix * p.out2.ld + c_x) = g;
                            store_data_settle();
                        }
                    } else {                                                                                                          // FIERY_EPI_GRU_OUT
                        const float4 u = *reinterpret_cast<const float4*>(p.aux0.ptr + e_o * p.aux0.istride + pix * p.aux0.ld + co);
                        const float4 h = *reinterpret_cast<const float4*>(p.aux1.ptr + e_o * p.aux1.istride + pix * p.aux1.ld + co);
                        float4 hn;
                        { const float a1 = (1.0f - u.x) * h.x, b1 = u.x * fmaxf(v.x, 0.f); hn.x = a1 + b1; }
                        { const float a1 = (1.0f - u.y) * h.y, b1 = u.y * fmaxf(v.y, 0.f); hn.y = a1 + b1; }
                        { const float a1 = (1.0f - u.z) * h.z, b1 = u.z * fmaxf(v.z, 0.f); hn.z = a1 + b1; }
                        { const float a1 = (1.0f - u.w) * h.w, b1 = u.w * fmaxf(v.w, 0.f); hn.w = a1 + b1; }
                        *reinterpret_cast<float4*>(p.out.ptr + e_o * p.out.istride + pix * p.out.ld + co) = hn;
                        store_data_settle();
                        if (p.out2.ptr) *reinterpret_cast<float4*>(p.out2.ptr + e_o * p.out2.istride + pix * p.out2.ld + co) = hn;
                        store_data_settle();
                    }
                }
        }
        }
    };
    if (W_EXP == 4) {
        float live_ = 0.f;                                         // (every accumulator block stays alive: no MFMA may be optimised away)
#pragma unroll
        for (int a = 0; a < PP; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) live_ += acc[a][b][r];
        if (live_ == 1.234e-30f) p.out.ptr[tid] = 0.f;
        return;
    }
    epilogue_block(std::integral_constant<int, 0>{});
    __syncthreads();                                                // everyone has read the first block before the second overwrites it
    epilogue_block(std::integral_constant<int, 1>{});
#if W_TRACE
    if (trace) trace[3] = wall_clock64();
#endif
    if (heads && e_worker) {
#pragma unroll
        for (int o = 0; o < FIERY_MAX_HEAD_OUTPUTS; ++o)
            if (o < p.heads.n_out && p.heads.group[o] == tile_n) {                 // (wave-uniform: every lane takes part in the exchange)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = hp[o][q];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    const int y = e_y + (q >> 1), x = e_x + (q & 1);
                    if (cq == 0 && e_live && y < H && x < W) {
                        v += p.heads.bias[o];
                        if (p.heads.sigmoid[o]) v = sigmoidf(v);
                        p.heads.out[o][e_o * p.heads.istride[o] + static_cast<long long>(y) * W + x] = v;
                    }
                }
            }
    }
}

}  // namespace

#if FIERY_WINOGRAD_SPLIT
// floats' worth of the split image: three bf16 terms per transformed weight (6 bytes where the fp32 image has 4)
size_t conv_winograd_split_packed_floats(int cout, int cin_units) {
    return static_cast<size_t>(wino_packed_floats((cout + 63) / 64 * 64, cin_units * 8)) * 3 / 2;
}
int conv_winograd_split_pack(const float* w, int cout, int cin_total, const ChanInverse& inv, int cin_units, float* packed, hipStream_t stream) {
    const long long total = wino_packed_floats((cout + 63) / 64 * 64, cin_units * 8) * 3;      // bf16 elements
    hipLaunchKernelGGL(k_pack_winograd_split, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream, w, cout, cin_total, inv,
                       cin_units, total, reinterpret_cast<unsigned short*>(packed));
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
#else
size_t conv_winograd_packed_floats(int cout, int cin_units) {
    return static_cast<size_t>(wino_packed_floats((cout + 63) / 64 * 64, cin_units * 8));
}

int conv_winograd_pack(const float* w, int cout, int cin_total, const ChanInverse& inv, int cin_units, float* packed, hipStream_t stream) {
    const long long total = wino_packed_floats((cout + 63) / 64 * 64, cin_units * 8);
    hipLaunchKernelGGL(k_pack_winograd, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream, w, cout, cin_total, inv, cin_units,
                       total, packed);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

#endif

// p.w = the Winograd-packed weights; p.M etc. as for the direct form
#if FIERY_WINOGRAD_SPLIT
bool conv_launch_winograd_split(const ConvP& p, hipStream_t stream) {
#else
bool conv_launch_winograd(const ConvP& p, hipStream_t stream) {
#endif
    const int TH = (p.Hout + 1) / 2, TW = (p.Wout + 1) / 2;
    const long long tiles = static_cast<long long>(p.n_img) * TH * TW;
    const dim3 grid(static_cast<unsigned>((tiles + WT - 1) / WT), static_cast<unsigned>(p.cout_pad / WBN));
    const bool dense = (p.vec_epilogue & 4) != 0;          // every tensor within 31-bit byte offsets of its base, 16-byte rows (conv_run)
    int kind = -1;
    if (p.epi == FIERY_EPI_HEADS) kind = (p.act == FIERY_ACT_NONE || p.act == FIERY_ACT_RELU) ? 4 : -1;
    else if (dense && p.epi == FIERY_EPI_PLAIN && (p.act == FIERY_ACT_NONE || p.act == FIERY_ACT_RELU))
        kind = (!p.res.ptr && !p.img_bias) ? 0 : 1;
    else if (dense && p.epi == FIERY_EPI_GRU_GATES) kind = 2;
    else if (dense && p.epi == FIERY_EPI_GRU_OUT) kind = 3;
    if (const char* forced = getenv("FIERY_WINOGRAD_GENERAL_EPILOGUE")) if (atoi(forced) != 0) kind = -1;      // tests
    int waves = W_DEFAULT_WAVES;                            // (read per launch: tests and A/B runs switch it)
    if (const char* e = getenv("FIERY_WINOGRAD_WAVES")) waves = atoi(e) == 8 ? 8 : atoi(e) == 4 ? 4 : waves;
    if (kind < 0) waves = 4;                                // (the general fallback does not fit 128 registers)
#if FIERY_WINOGRAD_SPLIT
#define FIERY_WINO_LAUNCH(K_) hipLaunchKernelGGL((k_conv_winograd<K_, 4, true>), grid, dim3(256), 0, stream, p)
#else
#define FIERY_WINO_LAUNCH(K_)                                                                                  \
    do {                                                                                                       \
        if (waves == 8) hipLaunchKernelGGL((k_conv_winograd<K_, 8>), grid, dim3(512), 0, stream, p);            \
        else hipLaunchKernelGGL((k_conv_winograd<K_, 4>), grid, dim3(256), 0, stream, p);                       \
    } while (0)
#endif
    switch (kind) {
        case 0: FIERY_WINO_LAUNCH(0); break;
        case 1: FIERY_WINO_LAUNCH(1); break;
        case 2: FIERY_WINO_LAUNCH(2); break;
        case 3: FIERY_WINO_LAUNCH(3); break;
        case 4: FIERY_WINO_LAUNCH(4); break;
        default: FIERY_WINO_LAUNCH(-1); break;
    }
#undef FIERY_WINO_LAUNCH
    return true;
}

}  // namespace fiery
