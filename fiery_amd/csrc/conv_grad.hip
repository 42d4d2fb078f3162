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

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_conv_wgrad(const float* in, int in_ld, int64_t in_img_stride, int cin_units, const float* grad_out,
                                int g_ld, int64_t g_img_stride, int cout, int n_img, int Hin, int Win, int Hout, int Wout, int kH,
                                int kW, int stride, int padH, int padW, float* dw, fiery_stream_t stream) {
    FIERY_REQUIRE(in && grad_out && dw, "conv_wgrad: null pointer");
    FIERY_REQUIRE(cin_units > 0 && cout > 0 && n_img > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "conv_wgrad: bad shape");
    FIERY_REQUIRE(kH >= 1 && kW >= 1 && stride >= 1 && padH >= 0 && padW >= 0, "conv_wgrad: bad kernel geometry");
    FIERY_REQUIRE(in_ld >= cin_units * 8 && g_ld >= cout, "conv_wgrad: leading dimension smaller than the channel count");
    FIERY_REQUIRE((Hin + 2 * padH - kH) / stride + 1 == Hout && (Win + 2 * padW - kW) / stride + 1 == Wout,
                  "conv_wgrad: output size does not belong to this convolution");
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
