// The element-wise half of `SpatialGRU.gru_cell` (fiery/layers/temporal.py:49-62) for training, forward and backward, on
// pixel-major rows:
//   r  = sigmoid(conv_reset(.) + b_r)          rh = (1 - r) * h                    (the state the candidate convolution sees)
//   u  = sigmoid(conv_update(.) + b_u)         h' = (1 - u) * h + u * p            (p = the candidate, after BatchNorm + ReLU)
// Autograd runs this as ~9 element-wise kernels forward and ~15 backward per cell, most of them on operands with different
// strides (the slow non-vectorised path: 30-45 us each on a 2 x 200 x 200 x 64 map); here each direction of each half is one
// streaming pass with 16-byte accesses.  Rows are [pixel][channel] with `ld` floats between pixels (convolution outputs
// with padded rows are read in place); outputs are dense rows of C_store channels, channels C .. C_store written as zeros.
#include "common.h"

namespace fiery {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

struct Rows {
    const float* p;
    int ld;
};

__device__ __forceinline__ float4 ld4(Rows r, long long pix, int c) { return *reinterpret_cast<const float4*>(r.p + pix * r.ld + c); }

// one thread = one pixel x four channels; `groups` = C_store / 4
template <typename F>
__device__ __forceinline__ void for_my_group(long long P, int groups, F&& f) {
    const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
    const long long pix = i / groups;
    if (pix >= P) return;
    f(pix, static_cast<int>(i - pix * groups) * 4);
}

__global__ __launch_bounds__(kThreads) void k_gru_reset_fwd(Rows pre, const float* __restrict__ bias, Rows h, long long P, int C, int groups,
                                                            float* __restrict__ r_out, float* __restrict__ rh_out, int C_store) {
    for_my_group(P, groups, [&](long long pix, int c) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f), rh = r;
        if (c < C) {
            const float4 a = ld4(pre, pix, c), hv = ld4(h, pix, c), b = *reinterpret_cast<const float4*>(bias + c);
            r = make_float4(sigm(a.x + b.x), sigm(a.y + b.y), sigm(a.z + b.z), sigm(a.w + b.w));
            rh = make_float4((1.f - r.x) * hv.x, (1.f - r.y) * hv.y, (1.f - r.z) * hv.z, (1.f - r.w) * hv.w);
        }
        *reinterpret_cast<float4*>(r_out + pix * C_store + c) = r;
        *reinterpret_cast<float4*>(rh_out + pix * C_store + c) = rh;
    });
}

// d_rh -> d_pre = -d_rh * h * r (1 - r),  dh = d_rh * (1 - r)
__global__ __launch_bounds__(kThreads) void k_gru_reset_bwd(Rows d_rh, const float* __restrict__ r_saved, Rows h, long long P, int C, int groups,
                                                            float* __restrict__ d_pre, float* __restrict__ dh, int C_store) {
    for_my_group(P, groups, [&](long long pix, int c) {
        float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), dhv = dp;
        if (c < C) {
            const float4 g = ld4(d_rh, pix, c), hv = ld4(h, pix, c), r = *reinterpret_cast<const float4*>(r_saved + pix * C_store + c);
            dp = make_float4(-g.x * hv.x * r.x * (1.f - r.x), -g.y * hv.y * r.y * (1.f - r.y), -g.z * hv.z * r.z * (1.f - r.z),
                             -g.w * hv.w * r.w * (1.f - r.w));
            dhv = make_float4(g.x * (1.f - r.x), g.y * (1.f - r.y), g.z * (1.f - r.z), g.w * (1.f - r.w));
        }
        *reinterpret_cast<float4*>(d_pre + pix * C_store + c) = dp;
        *reinterpret_cast<float4*>(dh + pix * C_store + c) = dhv;
    });
}

__global__ __launch_bounds__(kThreads) void k_gru_out_fwd(Rows pre, const float* __restrict__ bias, Rows h, Rows cand, long long P, int C, int groups,
                                                          float* __restrict__ u_out, float* __restrict__ hn_out, int C_store) {
    for_my_group(P, groups, [&](long long pix, int c) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f), hn = u;
        if (c < C) {
            const float4 a = ld4(pre, pix, c), hv = ld4(h, pix, c), pv = ld4(cand, pix, c), b = *reinterpret_cast<const float4*>(bias + c);
            u = make_float4(sigm(a.x + b.x), sigm(a.y + b.y), sigm(a.z + b.z), sigm(a.w + b.w));
            hn = make_float4((1.f - u.x) * hv.x + u.x * pv.x, (1.f - u.y) * hv.y + u.y * pv.y, (1.f - u.z) * hv.z + u.z * pv.z,
                             (1.f - u.w) * hv.w + u.w * pv.w);
        }
        *reinterpret_cast<float4*>(u_out + pix * C_store + c) = u;
        *reinterpret_cast<float4*>(hn_out + pix * C_store + c) = hn;
    });
}

// d_hn -> d_pre = d_hn * (p - h) * u (1 - u),  dh = d_hn * (1 - u),  dp = d_hn * u
__global__ __launch_bounds__(kThreads) void k_gru_out_bwd(Rows d_hn, const float* __restrict__ u_saved, Rows h, Rows cand, long long P, int C,
                                                          int groups, float* __restrict__ d_pre, float* __restrict__ dh,
                                                          float* __restrict__ dcand, int C_store) {
    for_my_group(P, groups, [&](long long pix, int c) {
        float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), dhv = dp, dc = dp;
        if (c < C) {
            const float4 g = ld4(d_hn, pix, c), hv = ld4(h, pix, c), pv = ld4(cand, pix, c);
            const float4 u = *reinterpret_cast<const float4*>(u_saved + pix * C_store + c);
            dp = make_float4(g.x * (pv.x - hv.x) * u.x * (1.f - u.x), g.y * (pv.y - hv.y) * u.y * (1.f - u.y),
                             g.z * (pv.z - hv.z) * u.z * (1.f - u.z), g.w * (pv.w - hv.w) * u.w * (1.f - u.w));
            dhv = make_float4(g.x * (1.f - u.x), g.y * (1.f - u.y), g.z * (1.f - u.z), g.w * (1.f - u.w));
            dc = make_float4(g.x * u.x, g.y * u.y, g.z * u.z, g.w * u.w);
        }
        *reinterpret_cast<float4*>(d_pre + pix * C_store + c) = dp;
        *reinterpret_cast<float4*>(dh + pix * C_store + c) = dhv;
        *reinterpret_cast<float4*>(dcand + pix * C_store + c) = dc;
    });
}

inline bool rows_ok(const float* p, int ld, int C) { return p && ld >= C && ld % 4 == 0 && aligned16(p); }

}  // namespace
}  // namespace fiery

using namespace fiery;

#define GRU_COMMON_CHECKS(name)                                                                                                   \
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && C % 4 == 0 && C_store >= C && C_store % 4 == 0, name ": channels must be multiples of 4"); \
    const int groups = C_store / 4;                                                                                               \
    const long long total = static_cast<long long>(n_pixels) * groups;                                                            \
    const dim3 grid(ceil_div(total, kThreads)), block(kThreads);                                                                   \
    hipStream_t hs = as_stream(stream);

extern "C" int fiery_gru_reset_fwd(const float* pre, int pre_ld, const float* bias, const float* h, int h_ld, int64_t n_pixels, int C,
                                   float* r, float* rh, int C_store, fiery_stream_t stream) {
    FIERY_REQUIRE(rows_ok(pre, pre_ld, C) && rows_ok(h, h_ld, C) && bias && aligned16(bias) && r && rh && aligned16(r) && aligned16(rh),
                  "gru_reset_fwd: null or misaligned operand");
    GRU_COMMON_CHECKS("gru_reset_fwd")
    hipLaunchKernelGGL(k_gru_reset_fwd, grid, block, 0, hs, Rows{pre, pre_ld}, bias, Rows{h, h_ld}, static_cast<long long>(n_pixels), C, groups, r, rh,
                       C_store);
    return check_launch("gru_reset_fwd");
}

extern "C" int fiery_gru_reset_bwd(const float* d_rh, int g_ld, const float* r, const float* h, int h_ld, int64_t n_pixels, int C,
                                   float* d_pre, float* dh, int C_store, fiery_stream_t stream) {
    FIERY_REQUIRE(rows_ok(d_rh, g_ld, C) && rows_ok(h, h_ld, C) && r && aligned16(r) && d_pre && dh && aligned16(d_pre) && aligned16(dh),
                  "gru_reset_bwd: null or misaligned operand");
    GRU_COMMON_CHECKS("gru_reset_bwd")
    hipLaunchKernelGGL(k_gru_reset_bwd, grid, block, 0, hs, Rows{d_rh, g_ld}, r, Rows{h, h_ld}, static_cast<long long>(n_pixels), C, groups, d_pre, dh,
                       C_store);
    return check_launch("gru_reset_bwd");
}

extern "C" int fiery_gru_out_fwd(const float* pre, int pre_ld, const float* bias, const float* h, int h_ld, const float* cand, int cand_ld,
                                 int64_t n_pixels, int C, float* u, float* h_new, int C_store, fiery_stream_t stream) {
    FIERY_REQUIRE(rows_ok(pre, pre_ld, C) && rows_ok(h, h_ld, C) && rows_ok(cand, cand_ld, C) && bias && aligned16(bias) && u && h_new &&
                      aligned16(u) && aligned16(h_new), "gru_out_fwd: null or misaligned operand");
    GRU_COMMON_CHECKS("gru_out_fwd")
    hipLaunchKernelGGL(k_gru_out_fwd, grid, block, 0, hs, Rows{pre, pre_ld}, bias, Rows{h, h_ld}, Rows{cand, cand_ld}, static_cast<long long>(n_pixels), C,
                       groups, u, h_new, C_store);
    return check_launch("gru_out_fwd");
}

extern "C" int fiery_gru_out_bwd(const float* d_hn, int g_ld, const float* u, const float* h, int h_ld, const float* cand, int cand_ld,
                                 int64_t n_pixels, int C, float* d_pre, float* dh, float* dcand, int C_store, fiery_stream_t stream) {
    FIERY_REQUIRE(rows_ok(d_hn, g_ld, C) && rows_ok(h, h_ld, C) && rows_ok(cand, cand_ld, C) && u && aligned16(u) && d_pre && dh && dcand &&
                      aligned16(d_pre) && aligned16(dh) && aligned16(dcand), "gru_out_bwd: null or misaligned operand");
    GRU_COMMON_CHECKS("gru_out_bwd")
    hipLaunchKernelGGL(k_gru_out_bwd, grid, block, 0, hs, Rows{d_hn, g_ld}, u, Rows{h, h_ld}, Rows{cand, cand_ld}, static_cast<long long>(n_pixels), C,
                       groups, d_pre, dh, dcand, C_store);
    return check_launch("gru_out_bwd");
}
