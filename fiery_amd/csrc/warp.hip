// Ego-motion warping of BEV feature maps for gfx950.
//
// Replaces `cumulative_warp_features` / `warp_features` and the pose algebra they call
// (fiery/utils/geometry.py:82-157, 181-253): on the reference these are ~40 small ATen launches plus
// affine_grid + grid_sample per past frame.  Here: one tiny kernel turns the ego-motion vectors into
// sampling transforms, one HBM-bound kernel resamples every past frame and at the same time changes
// the layout from the pooling kernel's channel-planes (NCHW) to the pixel-major layout (NHWC) the
// implicit-GEMM convolutions read.
#include "common.h"

namespace fiery {
namespace {

constexpr int kMaxFrames = 16;

// ---------------------------------------------------------------------------------------------------------------------
// Rounding contract.  This file is compiled with -ffp-contract=off: every product and sum below rounds on its own unless it
// is written as fmaf(), and the fmaf()s are exactly the fused operations of the ATen CPU kernels the reference runs
// (torch 2.x x86 build; determined by bit-comparison, tests/test_kernels_sim_aux.py keeps checking it):
//   * torch.linspace(-1, 1, n): first half fma(step, i, -1), second half fma(-step, n-1-i, 1), step = 2/(n-1);
//     affine_grid(align_corners=False) scales it as (v * (n-1)) / n            (AffineGridGenerator.cpp)
//   * the base grid times theta^T is a BLAS (MKL sgemm) product, and MKL picks its kernel by the HOST CPU: on Intel parts
//     it fuses, k ascending - fma(1, t2, fma(y, t1, x*t0)) -, on AMD EPYC parts it rounds every product and sum on its own
//     - (x*t0 + y*t1) + t2.  Both were measured (tools/probe/aten_warp_probe.py); FIERY_WARP_FUSED_GRID_PRODUCT in the
//     `flags` argument selects the form, and the Python binding asks the local ATen which one it is.
//   * grid_sample un-normalises with fma(g + 1, size/2, -0.5), forms the weights as (1-tx)(1-ty) ... tx.ty and sums
//     fma(v_se, w_se, fma(v_sw, w_sw, fma(v_ne, w_ne, v_nw*w_nw))) with 0 for corners outside (GridSamplerKernel.cpp)
//   * 3x3 / 4x4 matrix products: ATen's small-matrix loop, products and sums rounded separately, k ascending
//   * cos / sin / atan2 of the pose algebra go through double precision and are rounded once: the CPU's vector libraries
//     (MKL VML, SLEEF) return the correctly rounded value for ~95 % / 99.8 % of arguments and its neighbour otherwise,
//     which no device code can predict - `host` transforms (fiery_amd.model.host_warp_transforms) close that last gap.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float cos_rn(float v) { return static_cast<float>(cos(static_cast<double>(v))); }
__device__ inline float sin_rn(float v) { return static_cast<float>(sin(static_cast<double>(v))); }

// element i of  linspace(-1, 1, n) * (n - 1) / n
__device__ inline float base_coord(int i, int n) {
    if (n <= 1) return 0.f;
    const float step = __fdiv_rn(2.0f, static_cast<float>(n - 1));      // (a plain '/' is not correctly rounded in device code)
    const float l = i < n / 2 ? fmaf(step, static_cast<float>(i), -1.0f) : fmaf(-step, static_cast<float>(n - 1 - i), 1.0f);
    return __fdiv_rn(l * static_cast<float>(n - 1), static_cast<float>(n));
}

// un-normalised sampling position of output pixel (x, y) under the 2x3 transform th
__device__ inline void sample_position(const float* th, int x, int y, int W, int H, bool fused, float& fx, float& fy) {
    const float xb = base_coord(x, W), yb = base_coord(y, H);
    const float gx = (fused ? fmaf(yb, th[1], xb * th[0]) : xb * th[0] + yb * th[1]) + th[2];
    const float gy = (fused ? fmaf(yb, th[4], xb * th[3]) : xb * th[3] + yb * th[4]) + th[5];
    fx = fmaf(gx + 1.0f, static_cast<float>(W) * 0.5f, -0.5f);
    fy = fmaf(gy + 1.0f, static_cast<float>(H) * 0.5f, -0.5f);
}

__device__ void mat4_mul(const float* a, const float* b, float* out) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc += a[i * 4 + k] * b[k * 4 + j];
            out[i * 4 + j] = acc;
        }
}

// pose_vec2mat (geometry.py:143-157) with euler2mat's R = Rx.Ry.Rz (geometry.py:109-140)
__device__ void pose_to_mat(const float* v, float* m) {
    const float cx = cos_rn(v[3]), sx = sin_rn(v[3]);
    const float cy = cos_rn(v[4]), sy = sin_rn(v[4]);
    const float cz = cos_rn(v[5]), sz = sin_rn(v[5]);
    const float X[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    const float Y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    const float Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    float XY[9], R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc += X[i * 3 + k] * Y[k * 3 + j];
            XY[i * 3 + j] = acc;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc += XY[i * 3 + k] * Z[k * 3 + j];
            R[i * 3 + j] = acc;
        }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) m[i * 4 + j] = R[i * 3 + j];
        m[i * 4 + 3] = v[i];
    }
    m[12] = m[13] = m[14] = 0.f;
    m[15] = 1.f;
}

__global__ void k_warp_params(const float* __restrict__ ego, int B, int S, float ext_x, float ext_y,
                              float* __restrict__ theta, float* __restrict__ ego_shifted) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (ego_shifted) {
        // the temporal model's ego-pose channels: frame s carries the motion that led to it (fiery.py:152-154)
        float* sh = ego_shifted + static_cast<long long>(b) * S * 6;
        const float* src = ego + static_cast<long long>(b) * S * 6;
        for (int i = 0; i < 6; ++i) sh[i] = 0.f;
        for (int i = 6; i < S * 6; ++i) sh[i] = src[i - 6];
    }
    float* th = theta + static_cast<long long>(b) * S * 6;
    // the present frame is never resampled (geometry.py:245)
    float* last = th + (S - 1) * 6;
    last[0] = 1.f; last[1] = 0.f; last[2] = 0.f; last[3] = 0.f; last[4] = 1.f; last[5] = 0.f;
    if (S == 1) return;
    float cum[16], next[16], step[16];
    pose_to_mat(ego + (static_cast<long long>(b) * S + (S - 2)) * 6, cum);
    for (int t = S - 2; t >= 0; --t) {
        // mat2pose_vec keeps (tx, ty) and rz = atan2(-M01, M00) (geometry.py:82-106); warp_features
        // uses exactly those three (geometry.py:192-215)
        const float rz = static_cast<float>(atan2(static_cast<double>(-cum[1]), static_cast<double>(cum[0])));
        const float c = cos_rn(rz), s = sin_rn(rz);
        float* o = th + t * 6;
        o[0] = c;  o[1] = -s;  o[2] = __fdiv_rn(cum[7], ext_y);
        o[3] = s;  o[4] = c;   o[5] = -__fdiv_rn(cum[3], ext_x);
        if (t > 0) {
            pose_to_mat(ego + (static_cast<long long>(b) * S + (t - 1)) * 6, step);
            mat4_mul(step, cum, next);
            for (int i = 0; i < 16; ++i) cum[i] = next[i];
        }
    }
}

// cumulative_warp_features_reverse (geometry.py:256-280): frame 0 is the reference frame; frame i is sampled with
// inverse(flow[0]) @ ... @ inverse(flow[i-1]), the inverse of a pose matrix being [R^T | -R^T t] (geometry.py:160-178)
__device__ void invert_pose(const float* m, float* out) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[i * 4 + j] = m[j * 4 + i];
        float acc = 0.f;
        for (int k = 0; k < 3; ++k) acc += m[k * 4 + i] * m[k * 4 + 3];
        out[i * 4 + 3] = -acc;
    }
    out[12] = out[13] = out[14] = 0.f;
    out[15] = 1.f;
}

__global__ void k_warp_params_reverse(const float* __restrict__ ego, int B, int S, float ext_x, float ext_y,
                                      float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* th = theta + static_cast<long long>(b) * S * 6;
    th[0] = 1.f; th[1] = 0.f; th[2] = 0.f; th[3] = 0.f; th[4] = 1.f; th[5] = 0.f;      // frame 0 is never resampled
    float cum[16], step[16], inv[16], next[16];
    for (int i = 1; i < S; ++i) {
        pose_to_mat(ego + (static_cast<long long>(b) * S + (i - 1)) * 6, step);
        invert_pose(step, inv);
        if (i == 1) {
            for (int k = 0; k < 16; ++k) cum[k] = inv[k];
        } else {
            mat4_mul(cum, inv, next);
            for (int k = 0; k < 16; ++k) cum[k] = next[k];
        }
        const float rz = static_cast<float>(atan2(static_cast<double>(-cum[1]), static_cast<double>(cum[0])));
        const float c = cos_rn(rz), s = sin_rn(rz);
        float* o = th + i * 6;
        o[0] = c;  o[1] = -s;  o[2] = __fdiv_rn(cum[7], ext_y);
        o[3] = s;  o[4] = c;   o[5] = -__fdiv_rn(cum[3], ext_x);
    }
}

// grid_sample(mode='nearest', padding_mode='zeros', align_corners=False) of channel planes: labels stay NCHW.
// One thread per output pixel walks the channels (1 .. 6 for the label tensors of trainer.py:133-191).
__global__ __launch_bounds__(256) void k_bev_warp_nearest(const float* __restrict__ in, const float* __restrict__ theta,
                                                          int C, int H, int W, float* __restrict__ out, long long total,
                                                          int flags) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = static_cast<int>(i % W);
    const int y = static_cast<int>((i / W) % H);
    const int img = static_cast<int>(i / (static_cast<long long>(W) * H));
    const float* th = theta + img * 6;
    float px, py;
    sample_position(th, x, y, W, H, (flags & FIERY_WARP_FUSED_GRID_PRODUCT) != 0, px, py);
    // nearest: the un-normalised coordinate rounded half to even (ATen uses nearbyint)
    const float fx = nearbyintf(px);
    const float fy = nearbyintf(py);
    const bool inside = fx >= 0.f && fx < static_cast<float>(W) && fy >= 0.f && fy < static_cast<float>(H);
    const long long plane = static_cast<long long>(H) * W;
    const float* src = in + static_cast<long long>(img) * C * plane + (inside ? static_cast<long long>(fy) * W + static_cast<long long>(fx) : 0);
    float* dst = out + static_cast<long long>(img) * C * plane + static_cast<long long>(y) * W + x;
    for (int c = 0; c < C; ++c) dst[c * plane] = inside ? src[c * plane] : 0.f;
}

constexpr int kWarpTile = 64;   // pixels per workgroup (one row segment)
constexpr int kMaxWarpImages = 256;

// per-image "copy, do not resample" flags, passed by value as a kernel argument
struct IdentityFlags {
    unsigned char v[kMaxWarpImages];
};

// grid: (ceil(W / 64), H, n_img); 256 threads = 4 wavefronts.  Phase 1: wavefront w samples channels
// w, w+4, ... with lanes along x (unit-stride reads of the NCHW planes for small rotations).  Phase 2:
// the 64-pixel x C tile leaves LDS pixel-major with lanes along channels (unit-stride NHWC writes).
__global__ __launch_bounds__(256) void k_bev_warp(const float* __restrict__ in, const float* __restrict__ theta,
                                                  IdentityFlags identity, int C, int H, int W,
                                                  float* __restrict__ out, int out_ld, long long out_img_stride, int flags) {
    HIP_DYNAMIC_SHARED(float, tile)            // [kWarpTile][C + 1]
    const int img = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * kWarpTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int row = C + 1;
    const float* plane0 = in + static_cast<long long>(img) * C * H * W;
    const bool copy = identity.v[img] != 0;

    int ix0 = 0, iy0 = 0;
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
    bool v00 = false, v01 = false, v10 = false, v11 = false;
    if (x < W && !copy) {
        const float* th = theta + img * 6;
        // affine_grid + grid_sample un-normalisation, align_corners=False, in ATen's rounding order (top of file)
        float fx, fy;
        sample_position(th, x, y, W, H, (flags & FIERY_WARP_FUSED_GRID_PRODUCT) != 0, fx, fy);
        const float flx = floorf(fx), fly = floorf(fy);
        ix0 = static_cast<int>(flx);
        iy0 = static_cast<int>(fly);
        const float tx = fx - flx, ty = fy - fly;
        const float ex = 1.f - tx, sy = 1.f - ty;
        w00 = sy * ex;  w01 = sy * tx;
        w10 = ty * ex;  w11 = ty * tx;
        const bool xin0 = ix0 >= 0 && ix0 < W, xin1 = ix0 + 1 >= 0 && ix0 + 1 < W;
        const bool yin0 = iy0 >= 0 && iy0 < H, yin1 = iy0 + 1 >= 0 && iy0 + 1 < H;
        v00 = xin0 && yin0;  v01 = xin1 && yin0;  v10 = xin0 && yin1;  v11 = xin1 && yin1;
    }
    for (int c = wave; c < C; c += 4) {
        float val = 0.f;
        if (x < W) {
            const float* pl = plane0 + static_cast<long long>(c) * H * W;
            if (copy) {
                val = pl[y * W + x];
            } else {
                // zeros padding: out-of-range corners enter as 0; the sum is ATen's fused chain
                const float p00 = v00 ? pl[iy0 * W + ix0] : 0.f;
                const float p01 = v01 ? pl[iy0 * W + ix0 + 1] : 0.f;
                const float p10 = v10 ? pl[(iy0 + 1) * W + ix0] : 0.f;
                const float p11 = v11 ? pl[(iy0 + 1) * W + ix0 + 1] : 0.f;
                val = fmaf(p11, w11, fmaf(p10, w10, fmaf(p01, w01, p00 * w00)));
            }
        }
        tile[lane * row + c] = val;
    }
    __syncthreads();
    float* obase = out + static_cast<long long>(img) * out_img_stride + (static_cast<long long>(y) * W + x0) * out_ld;
    const int npx = min(kWarpTile, W - x0);
    for (int i = threadIdx.x; i < npx * C; i += blockDim.x) {
        const int px = i / C, c = i - px * C;
        obase[static_cast<long long>(px) * out_ld + c] = tile[px * row + c];
    }
}

// Adjoint of k_bev_warp in its input.  Same grid and tile; phase 1: the 64-pixel x C tile of the output gradient arrives
// pixel-major with lanes along channels; phase 2: wavefront w scatters channels w, w + 4, ... with lanes along x - the four
// corners of neighbouring pixels are neighbouring addresses of one channel plane for small rotations.
__global__ __launch_bounds__(256) void k_bev_warp_bwd(const float* __restrict__ gy, int g_ld, long long g_img_stride,
                                                      const float* __restrict__ theta, IdentityFlags identity, int C, int H, int W,
                                                      float* __restrict__ gx, int flags) {
    HIP_DYNAMIC_SHARED(float, tile)            // [kWarpTile][C + 1]
    const int img = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * kWarpTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int row = C + 1;
    const float* gbase = gy + static_cast<long long>(img) * g_img_stride + (static_cast<long long>(y) * W + x0) * g_ld;
    const int npx = min(kWarpTile, W - x0);
    for (int i = threadIdx.x; i < npx * C; i += blockDim.x) {
        const int px = i / C, c = i - px * C;
        tile[px * row + c] = gbase[static_cast<long long>(px) * g_ld + c];
    }
    __syncthreads();
    if (x >= W) return;
    float* plane0 = gx + static_cast<long long>(img) * C * H * W;
    if (identity.v[img] != 0) {                                   // copied, not resampled: every pixel is written exactly once
        for (int c = wave; c < C; c += 4) plane0[static_cast<long long>(c) * H * W + y * W + x] = tile[lane * row + c];
        return;
    }
    float fx, fy;
    sample_position(theta + img * 6, x, y, W, H, (flags & FIERY_WARP_FUSED_GRID_PRODUCT) != 0, fx, fy);
    const float flx = floorf(fx), fly = floorf(fy);
    const int ix0 = static_cast<int>(flx), iy0 = static_cast<int>(fly);
    const float tx = fx - flx, ty = fy - fly;
    const float ex = 1.f - tx, sy = 1.f - ty;
    const float w00 = sy * ex, w01 = sy * tx, w10 = ty * ex, w11 = ty * tx;
    const bool xin0 = ix0 >= 0 && ix0 < W, xin1 = ix0 + 1 >= 0 && ix0 + 1 < W;
    const bool yin0 = iy0 >= 0 && iy0 < H, yin1 = iy0 + 1 >= 0 && iy0 + 1 < H;
    for (int c = wave; c < C; c += 4) {
        const float g = tile[lane * row + c];
        float* pl = plane0 + static_cast<long long>(c) * H * W;
        if (xin0 && yin0) atomicAdd(pl + iy0 * W + ix0, g * w00);
        if (xin1 && yin0) atomicAdd(pl + iy0 * W + ix0 + 1, g * w01);
        if (xin0 && yin1) atomicAdd(pl + (iy0 + 1) * W + ix0, g * w10);
        if (xin1 && yin1) atomicAdd(pl + (iy0 + 1) * W + ix0 + 1, g * w11);
    }
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_bev_warp_bwd_nhwc_to_nchw(const float* grad_out, int g_ld, int64_t g_img_stride, const float* theta,
                                               const uint8_t* identity, int n_img, int C, int H, int W, float* grad_in, int flags,
                                               fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && theta && grad_in, "bev_warp_bwd: null pointer");
    FIERY_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0 && g_ld >= C, "bev_warp_bwd: bad shape");
    FIERY_REQUIRE(static_cast<size_t>(kWarpTile) * (C + 1) * sizeof(float) <= 160 * 1024, "bev_warp_bwd: too many channels");
    const size_t image = static_cast<size_t>(C) * H * W;
    if (hipMemsetAsync(grad_in, 0, image * n_img * sizeof(float), as_stream(stream)) != hipSuccess)
        return fail(FIERY_ELAUNCH, "bev_warp_bwd: cannot clear the input gradient");
    for (int i0 = 0; i0 < n_img; i0 += kMaxWarpImages) {
        const int n = n_img - i0 < kMaxWarpImages ? n_img - i0 : kMaxWarpImages;
        IdentityFlags ident;
        for (int i = 0; i < kMaxWarpImages; ++i) ident.v[i] = (identity && i < n && identity[i0 + i]) ? 1 : 0;
        hipLaunchKernelGGL(k_bev_warp_bwd, dim3(ceil_div(W, kWarpTile), H, n), dim3(256),
                           static_cast<size_t>(kWarpTile) * (C + 1) * sizeof(float), as_stream(stream),
                           grad_out + static_cast<long long>(i0) * g_img_stride, g_ld, static_cast<long long>(g_img_stride),
                           theta + static_cast<long long>(i0) * 6, ident, C, H, W, grad_in + static_cast<long long>(i0) * image, flags);
        int rc = check_launch("bev_warp_bwd");
        if (rc) return rc;
    }
    return FIERY_OK;
}

extern "C" int fiery_warp_params(const float* future_egomotion, int B, int S, float extent_x, float extent_y,
                                 float* theta, float* ego_shifted, fiery_stream_t stream) {
    FIERY_REQUIRE(future_egomotion && theta && B > 0 && S > 0, "warp_params: bad argument");
    FIERY_REQUIRE(S <= kMaxFrames, "warp_params: at most %d frames", kMaxFrames);
    FIERY_REQUIRE(extent_x != 0.f && extent_y != 0.f, "warp_params: zero spatial extent");
    hipLaunchKernelGGL(k_warp_params, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), future_egomotion, B, S,
                       extent_x, extent_y, theta, ego_shifted);
    return check_launch("warp_params");
}

extern "C" int fiery_bev_warp_nchw_to_nhwc(const float* in, const float* theta, const uint8_t* identity, int n_img, int C,
                                           int H, int W, float* out, int out_ld, int64_t out_img_stride, int flags,
                                           fiery_stream_t stream) {
    FIERY_REQUIRE(in && theta && out, "bev_warp: null pointer");
    FIERY_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0 && out_ld >= C, "bev_warp: bad shape");
    FIERY_REQUIRE(static_cast<size_t>(kWarpTile) * (C + 1) * sizeof(float) <= 160 * 1024, "bev_warp: too many channels");
    for (int i0 = 0; i0 < n_img; i0 += kMaxWarpImages) {
        const int n = n_img - i0 < kMaxWarpImages ? n_img - i0 : kMaxWarpImages;
        IdentityFlags ident;
        for (int i = 0; i < kMaxWarpImages; ++i) ident.v[i] = (identity && i < n && identity[i0 + i]) ? 1 : 0;
        hipLaunchKernelGGL(k_bev_warp, dim3(ceil_div(W, kWarpTile), H, n), dim3(256),
                           static_cast<size_t>(kWarpTile) * (C + 1) * sizeof(float), as_stream(stream),
                           in + static_cast<long long>(i0) * C * H * W, theta + static_cast<long long>(i0) * 6, ident, C, H, W,
                           out + static_cast<long long>(i0) * out_img_stride, out_ld, static_cast<long long>(out_img_stride), flags);
        int rc = check_launch("bev_warp");
        if (rc) return rc;
    }
    return FIERY_OK;
}

extern "C" int fiery_warp_params_reverse(const float* future_egomotion, int B, int S, float extent_x, float extent_y,
                                         float* theta, fiery_stream_t stream) {
    FIERY_REQUIRE(future_egomotion && theta && B > 0 && S > 0, "warp_params_reverse: bad argument");
    FIERY_REQUIRE(extent_x != 0.f && extent_y != 0.f, "warp_params_reverse: zero spatial extent");
    hipLaunchKernelGGL(k_warp_params_reverse, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), future_egomotion, B, S,
                       extent_x, extent_y, theta);
    return check_launch("warp_params_reverse");
}

extern "C" int fiery_bev_warp_nearest_nchw(const float* in, const float* theta, int n_img, int C, int H, int W, float* out,
                                           int flags, fiery_stream_t stream) {
    FIERY_REQUIRE(in && theta && out && in != out, "bev_warp_nearest: null pointer or in-place call");
    FIERY_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0, "bev_warp_nearest: bad shape");
    const long long total = static_cast<long long>(n_img) * H * W;
    hipLaunchKernelGGL(k_bev_warp_nearest, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), in, theta, C, H, W, out,
                       total, flags);
    return check_launch("bev_warp_nearest");
}
