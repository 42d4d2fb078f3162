// Weight gradient of the implicit-GEMM convolution (training).
//
// Replaces what autograd runs for `conv2d`'s weight in the reference's BEV stack (cuDNN / MIOpen wgrad behind
// fiery/layers/convolutions.py:9-168, fiery/layers/temporal.py:10-62, fiery/models/decoder.py:53-91):
//   dW[cout][tap][c] = sum over output pixels p of  dY[p][cout] * X[p * stride + tap - pad][c]
// - a GEMM whose reduction dimension is the PIXELS (120,000 per GRU layer) and whose result is small (cout x taps x cin).
// Mapping to CDNA4: v_mfma_f32_32x32x2_f32 with k = two neighbouring output pixels of a row; a wavefront owns a
// (64 couts) x (one tap, 64 input channels) block of dW over a share of the output rows and hands it to HBM with fp32
// atomics (the caller zeroes dW; the order of those additions moves the last bits from run to run).  Both operands are
// read pixel-major (NHWC) straight from global memory: a lane's two couts (or channels) of a pixel are 128 bytes apart in
// the same row, so a wavefront load is two full 128-byte lines per pixel.  At 16 flop per loaded byte the loop would need
// ~10 TB/s from L2 to keep the matrix cores busy, so the four wavefronts of a workgroup take NEIGHBOURING blocks - the
// taps of one (cout tile, channel tile) - over the SAME rows: they read the same dY lines and overlapping X lines within
// a few hundred cycles of each other, and three of four requests are served by the CU's vector cache.
// (The data gradient reuses the forward kernel with transposed, mirrored weights: fiery_amd/train_graph.py.)
#include "common.h"
#include <fiery_gfx950.h>

#include <cstdlib>

namespace fiery {
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

struct WgradP {
    const float* x;
    const float* g;
    float* dw;
    long long x_istride, g_istride;
    int x_ld, g_ld, cin_pad, cout;
    int n_img, Hin, Win, Hout, Wout, kH, kW, stride, padH, padW;
    int c_tiles, co_tiles;  // 64-channel tiles per tap, 64-cout tiles
};

// blocks q = (cout tile, channel tile, tap), taps fastest; grid (ceil(Q / per_wg), 1, row shares); 256 threads.
// per_wg = min(4, Q) blocks per workgroup, one per wavefront; with fewer than four blocks the spare wavefronts split rows.
__global__ __launch_bounds__(256) void k_conv_wgrad(WgradP p) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, kk = lane >> 5;
    const int taps = p.kH * p.kW;
    const int Q = p.co_tiles * p.c_tiles * taps;
    const int per_wg = Q < 4 ? Q : 4;
    const int n_sub = 4 / per_wg;                            // row sub-shares inside the workgroup
    const int slot = wave % per_wg, sub = wave / per_wg;
    const int q = blockIdx.x * per_wg + slot;
    if (q >= Q || sub >= n_sub) return;
    const int tap = q % taps, ct = (q / taps) % p.c_tiles, cot = q / (taps * p.c_tiles);
    const int co0 = cot * 64, c0 = ct * 64;
    const int dy = tap / p.kW, dx = tap - dy * p.kW;
    const int n_rows = p.n_img * p.Hout;
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool co_ok[2] = {co0 + m < p.cout, co0 + 32 + m < p.cout};
    const bool c_ok[2] = {c0 + m < p.cin_pad, c0 + 32 + m < p.cin_pad};
    const int row_step = gridDim.z * n_sub;
    for (int row = blockIdx.z * n_sub + sub; row < n_rows; row += row_step) {
        const int img = row / p.Hout, y = row - img * p.Hout;
        const int iy = y * p.stride + dy - p.padH;
        if (iy < 0 || iy >= p.Hin) continue;                // the tap looks at the zero padding: nothing to add
        // the two rows as buffers of their own (wave-uniform bases, 32-bit lane offsets): a lane whose pixel, cout or channel
        // is outside gets an offset past the descriptor and reads 0 - no branch around any load, so the loads of a trip are a
        // fixed count the hardware counter (vmcnt) can be waited on partially
        const __amdgpu_buffer_rsrc_t grow = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.g + img * p.g_istride + static_cast<long long>(y) * p.Wout * p.g_ld), 0, p.Wout * p.g_ld * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrow = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + img * p.x_istride + static_cast<long long>(iy) * p.Win * p.x_ld), 0, p.Win * p.x_ld * 4, 0x00020000);
        constexpr int kOutside = static_cast<int>(0x80000000u);
        const int g_lane[2] = {co_ok[0] ? (co0 + m) * 4 : kOutside, co_ok[1] ? (co0 + 32 + m) * 4 : kOutside};
        const int x_lane[2] = {c_ok[0] ? (c0 + m) * 4 : kOutside, c_ok[1] ? (c0 + 32 + m) * 4 : kOutside};
        // two pixel pairs per trip, and the operands of trip i + 1 are requested before the eight MFMAs of trip i: with
        // three wavefronts per SIMD a wavefront has to cover most of its own memory latency
        float a[2][2], b[2][2];
        auto request = [&](int x0, float (&fa)[2][2], float (&fb)[2][2]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int x = x0 + 2 * h + kk;                  // this lane's pixel of the pair
                const int ix = x * p.stride + dx - p.padW;
                const bool px_ok = x < p.Wout;
                const bool in_ok = px_ok && ix >= 0 && ix < p.Win;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int go = (px_ok && g_lane[t] != kOutside) ? x * p.g_ld * 4 + g_lane[t] : kOutside;
                    const int xo = (in_ok && x_lane[t] != kOutside) ? ix * p.x_ld * 4 + x_lane[t] : kOutside;
                    fa[h][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grow, go, 0, 0));
                    fb[h][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrow, xo, 0, 0));
                }
            }
        };
        request(0, a, b);
        for (int x0 = 0; x0 < p.Wout; x0 += 4) {
            float na[2][2], nb[2][2];
            request(x0 + 4, na, nb);                             // past the row's end every predicate is false: zeros, no access
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[h][ta], b[h][tb], acc[ta][tb], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[h][t] = na[h][t];
                    b[h][t] = nb[h][t];
                }
        }
    }
    // the block goes out from the accumulators: for a fixed register the 32 lanes of a half hold 32 consecutive channels
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int c = c0 + 32 * tb + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const float v = acc[ta][tb][r];
                if (co < p.cout && c < p.cin_pad && v != 0.f) atomicAdd(&p.dw[(static_cast<long long>(co) * taps + tap) * p.cin_pad + c], v);
            }
        }
}

// ---- 3 x 3, stride 1 (87 % of a training step's weight-gradient work): operands staged once in LDS for all nine taps -----------
// The kernel above reads dY and X from L2 once per tap.  Here a workgroup owns a (64 couts) x (64 channels) block of dW for
// ALL nine taps over a column strip of the map - `seg` (<= 56) output pixels wide, a range of output rows - and walks down the
// strip: per output row it stages that row's dY segment and ONE new X row (the other two of the 3-row window are already in a
// ring in LDS), so every dY and X element is fetched from global memory once per (cout tile, channel tile) instead of nine
// times.  Each wavefront owns a 32 x 32 quarter of the block for the nine taps (144 accumulator registers) and reads its
// operands from LDS with one ds_read_b32 per MFMA; when the layer is narrower than 64 couts or channels the spare wavefronts
// split the segment's pixels instead.  The next row's global loads are issued before the current row's 288 MFMAs and written
// to LDS after them (register staging: 33 KB per workgroup and step).
constexpr int kSegMax = 56;

struct Wgrad3P {
    const float* x;
    const float* g;
    float* dw;
    long long x_istride, g_istride;
    int x_ld, g_ld, cin_pad, cout;
    int n_img, H, W;            // 3 x 3, stride 1, pad 1: input and output maps have the same size
    int c_tiles, co_tiles, seg, n_seg, rows_per_wg, row_parts;
};

// BF16 (mixed-precision training, fiery_conv_wgrad_prec): the K dimension - output pixels - goes through
// v_mfma_f32_32x32x16_bf16, sixteen pixels per instruction instead of two.  The operands stay fp32 in LDS in the same
// [pixel][channel] image; a lane's eight pixels of one cout (channel) are eight ds_read_b32 of one bank column, rounded to
// bf16 in registers (round to nearest even), and the ten input pixels a lane needs for the three horizontal taps are read
// once: taps dx = 0 and dx = 2 share their bf16 pairs, dx = 1 pairs them the other way.  Per sixteen pixels and wavefront:
// 38 LDS reads, 31 packed conversions, 9 MFMAs of 32 cycles where the fp32 form issues 72 of 64.  Products of bf16 values
// are exact in fp32 and the accumulation is fp32: the result is the weight gradient of the ROUNDED dY and X.
// SPLIT (round 6, FIERY_PRECISION_F32_SPLIT): the same loop with every fp32 operand as three bf16 terms (x = x1 + x2 + x3 exactly)
// and six partial products per product, smallest first - fp32 accuracy (not below the fp32 instruction's: tools/probe/
// split_bf16_probe.hip) at 54 short MFMAs per sixteen pixels where the fp32 form issues 72 long ones.  Both operands are
// activations: each is split by the lane that read it (the ten input pixels of a row once, for all three taps).
template <bool BF16, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad3x3(Wgrad3P p) {
    static_assert(!SPLIT || BF16, "the split form is a mode of the bf16 loop");
    constexpr int G_ROWS = BF16 ? 64 : kSegMax;               // (bf16: whole 16-pixel steps; rows past the segment hold zeros)
    constexpr int X_ROWS = BF16 ? 66 : kSegMax + 2;
    __shared__ float s_g[G_ROWS * 64];                        // dY segment [pixel][cout]
    __shared__ float s_x[3][X_ROWS * 64];                     // ring of X rows [pixel + 1][channel]; ring slot = input row % 3
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kk = lane >> 5;
    // which strip: blockIdx.x = ((co tile * c_tiles + c tile) * n_seg + segment), blockIdx.y = (image, row part)
    int b = blockIdx.x;
    const int sg = b % p.n_seg;  b /= p.n_seg;
    const int ct = b % p.c_tiles, cot = b / p.c_tiles;
    const int img = blockIdx.y / p.row_parts, part = blockIdx.y - img * p.row_parts;
    const int y0 = part * p.rows_per_wg;
    const int y1 = min(y0 + p.rows_per_wg, p.H);
    const int x0 = sg * p.seg;
    const int co0 = cot * 64, c0 = ct * 64;
    // wavefront roles: 32-wide sub-blocks that exist, spare wavefronts split the K-steps
    const int n_cow = (p.cout - co0 > 32) ? 2 : 1, n_cw = (p.cin_pad - c0 > 32) ? 2 : 1;
    const int k_parts = 4 / (n_cow * n_cw);
    const int cw = wave % n_cw, cow = (wave / n_cw) % n_cow, k_part = wave / (n_cw * n_cow);
    // staging roles: thread -> (pixel slot, 16-byte channel group); 16 groups per pixel
    const int q = tid & 15, prow = tid >> 4;                  // 16 pixel slots per pass, 16 passes cover 256 pixels
    const bool g_q_ok = co0 + 4 * q + 3 < p.g_ld, x_q_ok = c0 + 4 * q + 3 < p.x_ld;
    constexpr int kOutside = static_cast<int>(0x80000000u);
    constexpr int G_SLOTS = (G_ROWS + 15) / 16, X_SLOTS = (X_ROWS + 15) / 16;
    float4 g_reg[G_SLOTS], x_reg[X_SLOTS];
    const float* g_img = p.g + img * p.g_istride;
    const float* x_img = p.x + img * p.x_istride;
    auto request_g = [&](int y) {                              // dY row y, pixels x0 .. x0 + seg - 1
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g_img + static_cast<long long>(y) * p.W * p.g_ld), 0, p.W * p.g_ld * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < G_SLOTS; ++i) {
            const int px = prow + 16 * i;
            const bool ok = g_q_ok && px < p.seg && x0 + px < p.W && y < p.H;
            const int off = ok ? ((x0 + px) * p.g_ld + co0 + 4 * q) * 4 : kOutside;
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            __builtin_memcpy(&g_reg[i], &raw, 16);
        }
    };
    auto request_x = [&](int iy) {                             // X row iy, pixels x0 - 1 .. x0 + seg (zeros outside the map)
        const int row = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x_img + static_cast<long long>(row) * p.W * p.x_ld), 0, p.W * p.x_ld * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < X_SLOTS; ++i) {
            const int px = prow + 16 * i;                      // slot px holds input pixel x0 - 1 + px
            const int ix = x0 - 1 + px;
            const bool ok = x_q_ok && px < p.seg + 2 && ix >= 0 && ix < p.W && iy >= 0 && iy < p.H;
            const int off = ok ? (ix * p.x_ld + c0 + 4 * q) * 4 : kOutside;
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            __builtin_memcpy(&x_reg[i], &raw, 16);
        }
    };
    auto store_g = [&]() {
#pragma unroll
        for (int i = 0; i < G_SLOTS; ++i) {
            const int px = prow + 16 * i;
            if (px < G_ROWS) *reinterpret_cast<float4*>(&s_g[px * 64 + 4 * q]) = g_reg[i];
        }
    };
    auto store_x = [&](int iy) {
        float* dst = s_x[(iy + 3) % 3];
#pragma unroll
        for (int i = 0; i < X_SLOTS; ++i) {
            const int px = prow + 16 * i;
            if (px < X_ROWS) *reinterpret_cast<float4*>(&dst[px * 64 + 4 * q]) = x_reg[i];
        }
    };
    v16f acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (y0 < y1) {
        // prologue: rows y0 - 1 and y0 of X into the ring, then the first step's operands into registers
        request_x(y0 - 1);
        store_x(y0 - 1);
        request_x(y0);
        store_x(y0);
        request_g(y0);
        request_x(y0 + 1);
        const int k_steps = (p.seg + 1) >> 1;
        for (int y = y0; y < y1; ++y) {
            __syncthreads();                                   // the previous row's readers are done with s_g and the oldest ring slot
            store_g();
            store_x(y + 1);
            __syncthreads();
            if (y + 1 < y1) {                                  // the next step's operands travel during this step's MFMAs
                request_g(y + 1);
                request_x(y + 2);
            }
            const float* xr0 = s_x[(y + 2) % 3];               // input rows y - 1, y, y + 1
            const float* xr1 = s_x[y % 3];
            const float* xr2 = s_x[(y + 1) % 3];
            const int a_off = cow * 32 + m, b_off = cw * 32 + m;
            if constexpr (BF16) {
                const int k_steps16 = (p.seg + 15) >> 4;
                for (int j = k_part; j < k_steps16; j += k_parts) {
                    const int px = 16 * j + 8 * kk;                // this lane's eight output pixels: px .. px + 7
                    float av[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) av[i] = s_g[(px + i) * 64 + a_off];
                    if constexpr (SPLIT) {
                        bf16x8 a3[3];
                        split_bf16x8(av, a3[0], a3[1], a3[2]);
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const float* xr = r == 0 ? xr0 : r == 1 ? xr1 : xr2;
                            // the three terms of the row's ten pixels, one term at a time (ten values + ten remainders live)
                            float v[10], t[10];
#pragma unroll
                            for (int i = 0; i < 10; ++i) v[i] = xr[(px + i) * 64 + b_off];
                            bf16x8 b3[3][3];                        // [tap dx][term]
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
#pragma unroll
                                for (int i = 0; i < 10; ++i) {
                                    t[i] = e < 2 ? bf16_round(v[i]) : v[i];          // (the last remainder is rounded as it is packed)
                                    v[i] -= t[i];
                                }
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx)
                                    b3[dx][e] = pack_bf16x8(make_float4(t[dx], t[dx + 1], t[dx + 2], t[dx + 3]),
                                                            make_float4(t[dx + 4], t[dx + 5], t[dx + 6], t[dx + 7]));
                            }
                            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1
#pragma unroll
                            for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) acc[3 * r + dx] = mfma_bf16_32x32x16(a3[TA[k6]], b3[dx][TB[k6]], acc[3 * r + dx]);
                        }
                        continue;
                    }
                    const bf16x8 a8 = pack_bf16x8(make_float4(av[0], av[1], av[2], av[3]), make_float4(av[4], av[5], av[6], av[7]));
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float* xr = r == 0 ? xr0 : r == 1 ? xr1 : xr2;
                        float v[10];
#pragma unroll
                        for (int i = 0; i < 10; ++i) v[i] = xr[(px + i) * 64 + b_off];
                        const bf16x8 b0 = pack_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
                        const bf16x8 b1 = pack_bf16x8(make_float4(v[1], v[2], v[3], v[4]), make_float4(v[5], v[6], v[7], v[8]));
                        const bf16x8 b2 = pack_bf16x8(make_float4(v[2], v[3], v[4], v[5]), make_float4(v[6], v[7], v[8], v[9]));
                        acc[3 * r] = mfma_bf16_32x32x16(a8, b0, acc[3 * r]);
                        acc[3 * r + 1] = mfma_bf16_32x32x16(a8, b1, acc[3 * r + 1]);
                        acc[3 * r + 2] = mfma_bf16_32x32x16(a8, b2, acc[3 * r + 2]);
                    }
                }
            } else
            for (int j = k_part; j < k_steps; j += k_parts) {
                const int px = 2 * j + kk;                     // this lane's output pixel of the pair (segment-relative)
                const float a = s_g[px * 64 + a_off];
                float bv[9];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    bv[dx] = xr0[(px + dx) * 64 + b_off];
                    bv[3 + dx] = xr1[(px + dx) * 64 + b_off];
                    bv[6 + dx] = xr2[(px + dx) * 64 + b_off];
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
            }
        }
    }
    // out: for a fixed register the 32 lanes of a half hold 32 consecutive channels of one cout
    const int c = c0 + cw * 32 + m;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cow * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float v = acc[t][r];
            if (co < p.cout && c < p.cin_pad && v != 0.f) atomicAdd(&p.dw[(static_cast<long long>(co) * 9 + t) * p.cin_pad + c], v);
        }
}

// ---- 1 x 1, stride 1: a plain (cout x pixels) . (pixels x channels) product, memory-bound -------------------------------------
// No spatial structure: the map is a flat list of pixels.  A workgroup owns a 64 x 64 block of dW and walks over chunks of 96
// pixels; both operands of a chunk are staged in LDS with 16-byte loads (the first kernel reads them with 4-byte loads, which the
// memory system serves at a third of the rate), the next chunk's loads travel during the current chunk's MFMAs.
constexpr int kChunk = 96;

struct Wgrad1P {
    const float* x;
    const float* g;
    float* dw;
    long long pixels;
    int x_ld, g_ld, cin_pad, cout, c_tiles, co_tiles, parts;
};

__global__ __launch_bounds__(256, 2) void k_conv_wgrad1x1(Wgrad1P p) {
    __shared__ float s_g[kChunk * 64];
    __shared__ float s_x[kChunk * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kk = lane >> 5;
    const int ct = blockIdx.x % p.c_tiles, cot = blockIdx.x / p.c_tiles;
    const int co0 = cot * 64, c0 = ct * 64;
    const int n_cow = (p.cout - co0 > 32) ? 2 : 1, n_cw = (p.cin_pad - c0 > 32) ? 2 : 1;
    const int k_parts = 4 / (n_cow * n_cw);
    const int cw = wave % n_cw, cow = (wave / n_cw) % n_cow, k_part = wave / (n_cw * n_cow);
    const int q = tid & 15, prow = tid >> 4;
    const bool g_q_ok = co0 + 4 * q + 3 < p.g_ld, x_q_ok = c0 + 4 * q + 3 < p.x_ld;
    constexpr int SLOTS = kChunk / 16;
    float4 g_reg[SLOTS], x_reg[SLOTS];
    const long long n_chunks = (p.pixels + kChunk - 1) / kChunk;
    auto request = [&](long long chunk) {
        const long long p0 = chunk * kChunk;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const long long px = p0 + prow + 16 * i;
            const bool ok = px < p.pixels && chunk < n_chunks;
            g_reg[i] = (ok && g_q_ok) ? *reinterpret_cast<const float4*>(p.g + px * p.g_ld + co0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            x_reg[i] = (ok && x_q_ok) ? *reinterpret_cast<const float4*>(p.x + px * p.x_ld + c0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int a_off = cow * 32 + m, b_off = cw * 32 + m;
    long long chunk = blockIdx.y;
    if (chunk < n_chunks) request(chunk);
    for (; chunk < n_chunks; chunk += p.parts) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int px = prow + 16 * i;
            *reinterpret_cast<float4*>(&s_g[px * 64 + 4 * q]) = g_reg[i];
            *reinterpret_cast<float4*>(&s_x[px * 64 + 4 * q]) = x_reg[i];
        }
        __syncthreads();
        request(chunk + p.parts);                              // past the last chunk: zeros, no access
        for (int j = k_part; j < kChunk / 2; j += k_parts) {
            const int px = 2 * j + kk;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_g[px * 64 + a_off], s_x[px * 64 + b_off], acc, 0, 0, 0);
        }
    }
    const int c = c0 + cw * 32 + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + cow * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        const float v = acc[r];
        if (co < p.cout && c < p.cin_pad && v != 0.f) atomicAdd(&p.dw[static_cast<long long>(co) * p.cin_pad + c], v);
    }
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_conv_wgrad(const float* in, int in_ld, int64_t in_img_stride, int cin_units, const float* grad_out,
                                int g_ld, int64_t g_img_stride, int cout, int n_img, int Hin, int Win, int Hout, int Wout, int kH,
                                int kW, int stride, int padH, int padW, float* dw, fiery_stream_t stream) {
    return fiery_conv_wgrad_prec(in, in_ld, in_img_stride, cin_units, grad_out, g_ld, g_img_stride, cout, n_img, Hin, Win, Hout, Wout,
                                 kH, kW, stride, padH, padW, FIERY_PRECISION_F32, dw, stream);
}

extern "C" int fiery_conv_wgrad_prec(const float* in, int in_ld, int64_t in_img_stride, int cin_units, const float* grad_out,
                                     int g_ld, int64_t g_img_stride, int cout, int n_img, int Hin, int Win, int Hout, int Wout, int kH,
                                     int kW, int stride, int padH, int padW, int precision, float* dw, fiery_stream_t stream) {
    FIERY_REQUIRE(precision == FIERY_PRECISION_F32 || precision == FIERY_PRECISION_BF16 || precision == FIERY_PRECISION_F32_SPLIT,
                  "conv_wgrad: unknown precision %d", precision);
    FIERY_REQUIRE(in && grad_out && dw, "conv_wgrad: null pointer");
    FIERY_REQUIRE(cin_units > 0 && cout > 0 && n_img > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "conv_wgrad: bad shape");
    FIERY_REQUIRE(kH >= 1 && kW >= 1 && stride >= 1 && padH >= 0 && padW >= 0, "conv_wgrad: bad kernel geometry");
    FIERY_REQUIRE(in_ld >= cin_units * 8 && g_ld >= cout, "conv_wgrad: leading dimension smaller than the channel count");
    FIERY_REQUIRE((Hin + 2 * padH - kH) / stride + 1 == Hout && (Win + 2 * padW - kW) / stride + 1 == Wout,
                  "conv_wgrad: output size does not belong to this convolution");
    static const int staged = [] {
        const char* e = getenv("FIERY_WGRAD_STAGED");
        return e ? atoi(e) : 1;
    }();
    if (staged && kH == 1 && kW == 1 && stride == 1 && padH == 0 && padW == 0 && in_ld % 4 == 0 && g_ld % 4 == 0 && aligned16(in) &&
        aligned16(grad_out) && (in_img_stride <= 0 || in_img_stride == static_cast<long long>(Hin) * Win * in_ld) &&
        (g_img_stride <= 0 || g_img_stride == static_cast<long long>(Hout) * Wout * g_ld)) {
        Wgrad1P q;
        q.x = in;  q.g = grad_out;  q.dw = dw;
        q.pixels = static_cast<long long>(n_img) * Hout * Wout;
        q.x_ld = in_ld;  q.g_ld = g_ld;  q.cin_pad = cin_units * 8;  q.cout = cout;
        q.c_tiles = ceil_div(q.cin_pad, 64);
        q.co_tiles = ceil_div(cout, 64);
        const int tiles = q.c_tiles * q.co_tiles;
        const long long n_chunks = (q.pixels + kChunk - 1) / kChunk;
        long long parts = ceil_div(768, tiles);
        if (parts > (n_chunks + 1) / 2) parts = (n_chunks + 1) / 2;
        if (parts < 1) parts = 1;
        if (parts > 65535) parts = 65535;
        q.parts = static_cast<int>(parts);
        hipLaunchKernelGGL(k_conv_wgrad1x1, dim3(tiles, q.parts), dim3(256), 0, as_stream(stream), q);
        return check_launch("conv_wgrad (1x1 staged)");
    }
    if (staged && kH == 3 && kW == 3 && stride == 1 && padH == 1 && padW == 1 && in_ld % 4 == 0 && g_ld % 4 == 0 && aligned16(in) &&
        aligned16(grad_out) && static_cast<long long>(Win) * (in_ld > g_ld ? in_ld : g_ld) * 4 < (1ll << 31)) {
        Wgrad3P q;
        q.x = in;  q.g = grad_out;  q.dw = dw;
        q.x_istride = in_img_stride > 0 ? in_img_stride : static_cast<long long>(Hin) * Win * in_ld;
        q.g_istride = g_img_stride > 0 ? g_img_stride : static_cast<long long>(Hout) * Wout * g_ld;
        if (q.x_istride % 4 == 0 && q.g_istride % 4 == 0) {
            q.x_ld = in_ld;  q.g_ld = g_ld;  q.cin_pad = cin_units * 8;  q.cout = cout;
            q.n_img = n_img;  q.H = Hout;  q.W = Wout;
            q.c_tiles = ceil_div(q.cin_pad, 64);
            q.co_tiles = ceil_div(cout, 64);
            q.n_seg = ceil_div(Wout, kSegMax);
            q.seg = ceil_div(Wout, q.n_seg);
            q.seg += q.seg & 1;                                   // whole pixel pairs
            q.n_seg = ceil_div(Wout, q.seg);
            // column strips x row parts: about two workgroups per CU, no part shorter than four rows (two X rows are loaded
            // just to start a part)
            const int strips = q.co_tiles * q.c_tiles * q.n_seg * n_img;
            int parts = ceil_div(512, strips);
            if (parts > ceil_div(Hout, 4)) parts = ceil_div(Hout, 4);
            if (parts < 1) parts = 1;
            q.rows_per_wg = ceil_div(Hout, parts);
            q.row_parts = ceil_div(Hout, q.rows_per_wg);
            FIERY_REQUIRE(static_cast<long long>(n_img) * q.row_parts < 65536, "conv_wgrad: grid too large");
            const dim3 grid3(q.co_tiles * q.c_tiles * q.n_seg, n_img * q.row_parts);
            if (precision == FIERY_PRECISION_F32_SPLIT) hipLaunchKernelGGL((k_conv_wgrad3x3<true, true>), grid3, dim3(256), 0, as_stream(stream), q);
            else if (precision == FIERY_PRECISION_BF16) hipLaunchKernelGGL(k_conv_wgrad3x3<true>, grid3, dim3(256), 0, as_stream(stream), q);
            else hipLaunchKernelGGL(k_conv_wgrad3x3<false>, grid3, dim3(256), 0, as_stream(stream), q);
            return check_launch("conv_wgrad (3x3 staged)");
        }
    }
    WgradP p;
    p.x = in;  p.g = grad_out;  p.dw = dw;
    p.x_istride = in_img_stride > 0 ? in_img_stride : static_cast<long long>(Hin) * Win * in_ld;
    p.g_istride = g_img_stride > 0 ? g_img_stride : static_cast<long long>(Hout) * Wout * g_ld;
    p.x_ld = in_ld;  p.g_ld = g_ld;  p.cin_pad = cin_units * 8;  p.cout = cout;
    p.n_img = n_img;  p.Hin = Hin;  p.Win = Win;  p.Hout = Hout;  p.Wout = Wout;
    p.kH = kH;  p.kW = kW;  p.stride = stride;  p.padH = padH;  p.padW = padW;
    p.c_tiles = ceil_div(p.cin_pad, 64);
    p.co_tiles = ceil_div(cout, 64);
    const int n_rows = n_img * Hout;
    // enough row shares to fill the chip (256 CUs x a few workgroups) without leaving a wavefront fewer than two rows
    const long long Q = static_cast<long long>(p.co_tiles) * p.c_tiles * kH * kW;
    const int per_wg = Q < 4 ? static_cast<int>(Q) : 4;
    const int n_sub = 4 / per_wg;
    const int wgs_q = ceil_div(Q, per_wg);
    static const int target_wgs = [] {
        const char* e = getenv("FIERY_WGRAD_WGS");
        return e && atoi(e) > 0 ? atoi(e) : 2048;
    }();
    int shares = ceil_div(target_wgs, wgs_q);
    if (shares > ceil_div(n_rows, 2 * n_sub)) shares = ceil_div(n_rows, 2 * n_sub);
    if (shares < 1) shares = 1;
    FIERY_REQUIRE(wgs_q < (1 << 30) && shares < 65536, "conv_wgrad: grid too large");
    hipLaunchKernelGGL(k_conv_wgrad, dim3(wgs_q, 1, shares), dim3(256), 0, as_stream(stream), p);
    return check_launch("conv_wgrad");
}
