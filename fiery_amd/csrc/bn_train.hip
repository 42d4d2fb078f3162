// BatchNorm with batch statistics (+ fused ReLU), forward and backward, on pixel-major activations (training; SURVEY.md
// section 8f rank 2).
//
// Replaces what autograd runs for `nn.BatchNorm2d / BatchNorm3d (+ nn.ReLU)` of the reference's BEV stack in train() mode
// (fiery/layers/convolutions.py:27-34, 85-105; fiery/layers/temporal.py:77-84, 107-117; fiery/models/decoder.py:13-14 ...):
//   mean_c = E[x_c], var_c = E[(x_c - mean_c)^2] over all pixels of all images (biased, as torch normalises),
//   y = max(0, (x - mean) * rsqrt(var + eps) * gamma + beta),   running statistics updated with the unbiased variance,
//   backward:  g' = g * [y > 0];  dbeta = sum g';  dgamma = sum g' * xhat;
//              dx = gamma * invstd * (g' - dbeta / P - xhat * dgamma / P)          (P = number of pixels).
// All four passes stream the tensor once at HBM speed (a 6 x 200 x 200 x 64 map is 61 MB: ~12 us at 5 TB/s); the ATen / MIOpen
// path spends six kernels plus layout copies on the same work.  Rows are [pixel][channel] with `ld` floats between pixels,
// so a convolution output with padded rows is read in place.
//
// Per-channel sums are two-stage and deterministic: every workgroup reduces its pixels in registers and LDS and writes one
// partial row; a second kernel adds the partial rows in a fixed order (in double: the inputs are fp32 sums of ~2,000 values
// each, the totals feed a subtraction).  The variance is accumulated around a per-channel pivot (the first pixel's value),
// which keeps E[d^2] - E[d]^2 well conditioned when |mean| >> std.
#include "common.h"

namespace fiery {
namespace {

constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 512;          // partial rows (workspace contract: see fiery_bn_workspace_floats)

// thread layout: c4 = t % groups (a float4 of channels), r = t / groups (pixel lane); `rows` pixel lanes per block
struct BnShape {
    int groups, rows, blocks;
    long long pixels_per_block;
};

inline BnShape bn_shape(long long P, int C) {
    BnShape s;
    s.groups = (C + 3) / 4;
    s.rows = kBnThreads / s.groups;
    if (s.rows < 1) s.rows = 1;
    long long blocks = (P + static_cast<long long>(s.rows) * 8 - 1) / (static_cast<long long>(s.rows) * 8);   // >= 8 pixels per lane
    if (blocks > kBnMaxBlocks) blocks = kBnMaxBlocks;
    if (blocks < 1) blocks = 1;
    s.blocks = static_cast<int>(blocks);
    s.pixels_per_block = (P + blocks - 1) / blocks;
    return s;
}

__device__ __forceinline__ float4 load4(const float* p, int c, int C, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(p + c);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) v.x = p[c];
    if (c + 1 < C) v.y = p[c + 1];
    if (c + 2 < C) v.z = p[c + 2];
    if (c + 3 < C) v.w = p[c + 3];
    return v;
}

__device__ __forceinline__ void store4(float* p, int c, int C_store, bool vec, float4 v) {
    if (vec) {
        *reinterpret_cast<float4*>(p + c) = v;
        return;
    }
    if (c < C_store) p[c] = v.x;
    if (c + 1 < C_store) p[c + 1] = v.y;
    if (c + 2 < C_store) p[c + 2] = v.z;
    if (c + 3 < C_store) p[c + 3] = v.w;
}

// block-level sum of two float4 per thread over the pixel lanes; result valid in the threads with r == 0
__device__ __forceinline__ void reduce_rows(float4& a, float4& b, int groups, int rows, float* lds) {
    const int t = threadIdx.x;
    float4* la = reinterpret_cast<float4*>(lds);
    float4* lb = la + kBnThreads;
    la[t] = a;
    lb[t] = b;
    __syncthreads();
    if (t < groups) {
        for (int r = 1; r < rows; ++r) {
            const float4 va = la[t + r * groups], vb = lb[t + r * groups];
            a.x += va.x; a.y += va.y; a.z += va.z; a.w += va.w;
            b.x += vb.x; b.y += vb.y; b.z += vb.z; b.w += vb.w;
        }
    }
}

// pass 1 of the forward: partial[block][0][c] = sum (x - pivot), partial[block][1][c] = sum (x - pivot)^2
__global__ __launch_bounds__(kBnThreads) void k_bn_partial_stats(const float* __restrict__ x, int ld, long long P, int C, int vec,
                                                                 BnShape s, int Cp, float* __restrict__ partial) {
    __shared__ float lds[2 * kBnThreads * 4];
    const int t = threadIdx.x, c4 = t % s.groups, r = t / s.groups, c = c4 * 4;
    const bool live = r < s.rows;
    const float4 pivot = load4(x, c, C, vec);                        // pixel 0
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (live) {
        const long long p0 = blockIdx.x * s.pixels_per_block;
        long long p1 = p0 + s.pixels_per_block;
        if (p1 > P) p1 = P;
        for (long long p = p0 + r; p < p1; p += s.rows) {
            const float4 v = load4(x + p * ld, c, C, vec);
            const float dx = v.x - pivot.x, dy = v.y - pivot.y, dz = v.z - pivot.z, dw = v.w - pivot.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
    }
    reduce_rows(s1, s2, s.groups, s.rows, lds);
    if (t < s.groups) {
        float* row = partial + static_cast<long long>(blockIdx.x) * 2 * Cp;
        *reinterpret_cast<float4*>(row + c) = s1;
        *reinterpret_cast<float4*>(row + Cp + c) = s2;
    }
}

// sum of partial[b][which][c] over the partial rows b, by one 64-lane group per channel: lane l adds rows l, l + 64, ...
// in double, the 64 lane sums are added in lane order by lane 0 (fixed order: deterministic)
__device__ __forceinline__ void sum_partials(const float* __restrict__ partial, int blocks, int Cp, int c, int lane, int grp,
                                             double (*lds)[2][64], double& s1, double& s2) {
    double a = 0.0, b = 0.0;
    for (int r = lane; r < blocks; r += 64) {
        a += partial[static_cast<long long>(r) * 2 * Cp + c];
        b += partial[static_cast<long long>(r) * 2 * Cp + Cp + c];
    }
    lds[grp][0][lane] = a;
    lds[grp][1][lane] = b;
    __syncthreads();
    s1 = 0.0;
    s2 = 0.0;
    if (lane == 0)
        for (int l = 0; l < 64; ++l) {
            s1 += lds[grp][0][l];
            s2 += lds[grp][1][l];
        }
}

// pass 2: totals in a fixed order -> mean, invstd (and the running statistics, torch's update rule); 4 channels per block
__global__ __launch_bounds__(256) void k_bn_finish_stats(const float* __restrict__ x, const float* __restrict__ partial, int blocks,
                                                         int Cp, int C, long long P, float eps, float momentum,
                                                         float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float* __restrict__ var_out) {
    __shared__ double lds[4][2][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + grp;
    const int cc = c < C ? c : C - 1;                            // (spare groups repeat the last channel and drop the result)
    double s1, s2;
    sum_partials(partial, blocks, Cp, cc, lane, grp, lds, s1, s2);
    if (lane != 0 || c >= C) return;
    const double n = static_cast<double>(P);
    const double d = s1 / n;
    double var = s2 / n - d * d;
    if (var < 0.0) var = 0.0;
    const float m = static_cast<float>(static_cast<double>(x[c]) + d);
    mean[c] = m;
    if (invstd) invstd[c] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (var_out) var_out[c] = static_cast<float>(var);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var) {
        const double unbiased = P > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * static_cast<float>(unbiased);
    }
}

// eval mode: the running statistics in the form the other kernels take
__global__ void k_bn_running_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C, float eps,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// The element-wise passes' thread layout (round 5): c4 = t % groups (a float4 of the STORED channels), r = t / groups (pixel lane),
// `rows` = kBnThreads / groups pixel lanes per block; a block covers kBnElemPixels * rows consecutive pixels, a lane walks
// pixels p0 + r, p0 + r + rows, ...  The per-channel parameters are loaded once per thread (they used to be loaded per float4:
// 16-20 parameter loads beside one or three payload loads, and a 64-bit division - the launches ran at 2.1 TB/s), the payload
// loads of four pixels are issued together.
constexpr int kBnElemPixels = 8;
struct BnElemShape {
    int groups, rows;
    unsigned blocks;
};
inline BnElemShape bn_elem_shape(long long P, int C_store) {
    BnElemShape s;
    s.groups = (C_store + 3) / 4;
    s.rows = kBnThreads / s.groups;
    if (s.rows < 1) s.rows = 1;
    s.blocks = static_cast<unsigned>(ceil_div(P, static_cast<long long>(s.rows) * kBnElemPixels));
    return s;
}

// pass 3: y = act((x - mean) * invstd * gamma + beta); channels C .. C_store of y are written as zeros (row padding the
// next convolution reads)
__global__ __launch_bounds__(kBnThreads) void k_bn_apply(const float* __restrict__ x, int ld, long long P, int C, int vec,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                         float* __restrict__ y, int y_ld, int C_store, int y_vec, int groups, int rows) {
    const int t = threadIdx.x, c4 = t % groups, r = t / groups, c = c4 * 4;
    if (r >= rows) return;
    const long long p0 = static_cast<long long>(blockIdx.x) * rows * kBnElemPixels;
    float m[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = c + j < C;
        m[j] = in ? mean[c + j] : 0.f;
        is[j] = in ? invstd[c + j] : 0.f;
        ga[j] = in ? (gamma ? gamma[c + j] : 1.f) : 0.f;
        be[j] = in ? (beta ? beta[c + j] : 0.f) : 0.f;
    }
#pragma unroll
    for (int k0 = 0; k0 < kBnElemPixels; k0 += 4) {
        float4 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long p = p0 + static_cast<long long>(k0 + k) * rows + r;
            xv[k] = (p < P && c < C) ? load4(x + p * ld, c, C, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long p = p0 + static_cast<long long>(k0 + k) * rows + r;
            if (p >= P) continue;
            const float in[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
            float out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = 0.f;
                if (c + j < C) {
                    v = (in[j] - m[j]) * is[j] * ga[j] + be[j];
                    if (relu) v = fmaxf(v, 0.f);
                }
                out[j] = v;
            }
            store4(y + p * y_ld, c, C_store, y_vec, make_float4(out[0], out[1], out[2], out[3]));
        }
    }
}

// backward pass 1: partial[block][0][c] = sum g', partial[block][1][c] = sum g' * xhat    (g' = g where y > 0)
__global__ __launch_bounds__(kBnThreads) void k_bn_bwd_partial(const float* __restrict__ g, int g_ld, int g_vec,
                                                               const float* __restrict__ x, int ld, int vec,
                                                               const float* __restrict__ y, int y_ld, int y_vec, long long P, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               BnShape s, int Cp, float* __restrict__ partial) {
    __shared__ float lds[2 * kBnThreads * 4];
    const int t = threadIdx.x, c4 = t % s.groups, r = t / s.groups, c = c4 * 4;
    const bool live = r < s.rows;
    float m[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = c + j < C ? mean[c + j] : 0.f;
        is[j] = c + j < C ? invstd[c + j] : 0.f;
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (live) {
        const long long p0 = blockIdx.x * s.pixels_per_block;
        long long p1 = p0 + s.pixels_per_block;
        if (p1 > P) p1 = P;
        for (long long p = p0 + r; p < p1; p += s.rows) {
            float4 gv = load4(g + p * g_ld, c, C, g_vec);
            const float4 xv = load4(x + p * ld, c, C, vec);
            if (y) {
                const float4 yv = load4(y + p * y_ld, c, C, y_vec);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            s1.x += gv.x; s1.y += gv.y; s1.z += gv.z; s1.w += gv.w;
            s2.x += gv.x * ((xv.x - m[0]) * is[0]); s2.y += gv.y * ((xv.y - m[1]) * is[1]);
            s2.z += gv.z * ((xv.z - m[2]) * is[2]); s2.w += gv.w * ((xv.w - m[3]) * is[3]);
        }
    }
    reduce_rows(s1, s2, s.groups, s.rows, lds);
    if (t < s.groups) {
        float* row = partial + static_cast<long long>(blockIdx.x) * 2 * Cp;
        *reinterpret_cast<float4*>(row + c) = s1;
        *reinterpret_cast<float4*>(row + Cp + c) = s2;
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finish(const float* __restrict__ partial, int blocks, int Cp, int C,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double lds[4][2][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + grp;
    const int cc = c < C ? c : C - 1;
    double s1, s2;
    sum_partials(partial, blocks, Cp, cc, lane, grp, lds, s1, s2);
    if (lane != 0 || c >= C) return;
    dbeta[c] = static_cast<float>(s1);
    dgamma[c] = static_cast<float>(s2);
}

// backward pass 2: dx = gamma * invstd * (g' - [batch] (dbeta + xhat * dgamma) / P); channels C .. C_store as zeros
// (thread layout and walk: see k_bn_apply)
__global__ __launch_bounds__(kBnThreads) void k_bn_bwd_dx(const float* __restrict__ g, int g_ld, int g_vec, const float* __restrict__ x,
                                                          int ld, int vec, const float* __restrict__ y, int y_ld, int y_vec,
                                                          long long P, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                          int batch_stats, long long P_total, float* __restrict__ dx, int dx_ld,
                                                          int C_store, int dx_vec, int groups, int rows) {
    const int t = threadIdx.x, c4 = t % groups, r = t / groups, c = c4 * 4;
    if (r >= rows) return;
    const long long p0 = static_cast<long long>(blockIdx.x) * rows * kBnElemPixels;
    const float inv_p = 1.f / static_cast<float>(P_total);           // the statistics' pixel count (all processes' for SyncBatchNorm)
    float m[4], is[4], ga[4], dg[4], db[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = c + j < C;
        m[j] = in ? mean[c + j] : 0.f;
        is[j] = in ? invstd[c + j] : 0.f;
        ga[j] = in ? (gamma ? gamma[c + j] : 1.f) : 0.f;
        dg[j] = in && batch_stats ? dgamma[c + j] : 0.f;
        db[j] = in && batch_stats ? dbeta[c + j] : 0.f;
    }
#pragma unroll
    for (int k0 = 0; k0 < kBnElemPixels; k0 += 4) {
        float4 gq[4], xq[4], yq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long p = p0 + static_cast<long long>(k0 + k) * rows + r;
            const bool on = p < P && c < C;
            gq[k] = on ? load4(g + p * g_ld, c, C, g_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            xq[k] = on ? load4(x + p * ld, c, C, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            yq[k] = (on && y) ? load4(y + p * y_ld, c, C, y_vec) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long p = p0 + static_cast<long long>(k0 + k) * rows + r;
            if (p >= P) continue;
            const float gv[4] = {gq[k].x, gq[k].y, gq[k].z, gq[k].w}, xv[4] = {xq[k].x, xq[k].y, xq[k].z, xq[k].w},
                        yv[4] = {yq[k].x, yq[k].y, yq[k].z, yq[k].w};
            float out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = 0.f;
                if (c + j < C) {
                    const float gg = yv[j] > 0.f ? gv[j] : 0.f;
                    const float xhat = (xv[j] - m[j]) * is[j];
                    v = batch_stats ? ga[j] * is[j] * (gg - (db[j] + xhat * dg[j]) * inv_p) : ga[j] * is[j] * gg;
                }
                out[j] = v;
            }
            store4(dx + p * dx_ld, c, C_store, dx_vec, make_float4(out[0], out[1], out[2], out[3]));
        }
    }
}

inline bool rows_vec(const float* p, int ld, int C) { return C % 4 == 0 && ld % 4 == 0 && aligned16(p); }

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int64_t fiery_bn_workspace_floats(int C) {
    const int Cp = (C + 3) / 4 * 4;
    return static_cast<int64_t>(kBnMaxBlocks) * 2 * Cp;
}

extern "C" int fiery_bn_train_fwd(const float* x, int ld, int64_t n_pixels, int C, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int batch_stats, float momentum, float eps, int relu,
                                  float* y, int y_ld, int C_store, float* mean, float* invstd, float* workspace, fiery_stream_t stream) {
    FIERY_REQUIRE(x && y && mean && invstd && workspace, "bn_train_fwd: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && ld >= C && C_store >= C && y_ld >= C_store, "bn_train_fwd: bad shape");
    FIERY_REQUIRE(batch_stats || (running_mean && running_var), "bn_train_fwd: eval mode needs the running statistics");
    FIERY_REQUIRE(C_store <= 4 * kBnThreads, "bn_train_fwd: at most %d stored channels", 4 * kBnThreads);     // (the apply pass deals channel quads to a workgroup's threads)
    hipStream_t hs = as_stream(stream);
    const BnShape s = bn_shape(n_pixels, C);
    const int Cp = (C + 3) / 4 * 4;
    const int vec = rows_vec(x, ld, C) ? 1 : 0;
    if (batch_stats) {
        hipLaunchKernelGGL(k_bn_partial_stats, dim3(s.blocks), dim3(kBnThreads), 0, hs, x, ld, static_cast<long long>(n_pixels), C, vec, s,
                           Cp, workspace);
        hipLaunchKernelGGL(k_bn_finish_stats, dim3(ceil_div(C, 4)), dim3(256), 0, hs, x, workspace, s.blocks, Cp, C,
                           static_cast<long long>(n_pixels), eps, momentum, mean, invstd, running_mean, running_var,
                           static_cast<float*>(nullptr));
    } else {
        hipLaunchKernelGGL(k_bn_running_stats, dim3(ceil_div(C, 64)), dim3(64), 0, hs, running_mean, running_var, C, eps, mean, invstd);
    }
    const BnElemShape es = bn_elem_shape(n_pixels, C_store);
    const int y_vec = (C_store % 4 == 0 && y_ld % 4 == 0 && aligned16(y)) ? 1 : 0;
    hipLaunchKernelGGL(k_bn_apply, dim3(es.blocks), dim3(kBnThreads), 0, hs, x, ld, static_cast<long long>(n_pixels), C,
                       vec, mean, invstd, gamma, beta, relu, y, y_ld, C_store, y_vec, es.groups, es.rows);
    return check_launch("bn_train_fwd");
}

namespace {
int bn_bwd_reduce(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels, int C,
                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* workspace, hipStream_t hs) {
    const BnShape s = bn_shape(n_pixels, C);
    const int Cp = (C + 3) / 4 * 4;
    const int vec = rows_vec(x, ld, C) ? 1 : 0, g_vec = rows_vec(grad_out, g_ld, C) ? 1 : 0, y_vec = (y && rows_vec(y, y_ld, C)) ? 1 : 0;
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(s.blocks), dim3(kBnThreads), 0, hs, grad_out, g_ld, g_vec, x, ld, vec, y, y_ld, y_vec,
                       static_cast<long long>(n_pixels), C, mean, invstd, s, Cp, workspace);
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3(ceil_div(C, 4)), dim3(256), 0, hs, workspace, s.blocks, Cp, C, dgamma, dbeta);
    return check_launch("bn_train_bwd (sums)");
}

int bn_bwd_dx(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels, int C,
              const float* gamma, const float* mean, const float* invstd, const float* dgamma, const float* dbeta, int batch_stats,
              int64_t total_pixels, float* grad_in, int gi_ld, int C_store, hipStream_t hs) {
    const int vec = rows_vec(x, ld, C) ? 1 : 0, g_vec = rows_vec(grad_out, g_ld, C) ? 1 : 0, y_vec = (y && rows_vec(y, y_ld, C)) ? 1 : 0;
    const BnElemShape es = bn_elem_shape(n_pixels, C_store);
    const int dx_vec = (C_store % 4 == 0 && gi_ld % 4 == 0 && aligned16(grad_in)) ? 1 : 0;
    hipLaunchKernelGGL(k_bn_bwd_dx, dim3(es.blocks), dim3(kBnThreads), 0, hs, grad_out, g_ld, g_vec, x, ld, vec, y, y_ld,
                       y_vec, static_cast<long long>(n_pixels), C, mean, invstd, gamma, dgamma, dbeta, batch_stats,
                       static_cast<long long>(total_pixels), grad_in, gi_ld, C_store, dx_vec, es.groups, es.rows);
    return check_launch("bn_train_bwd (dx)");
}
}  // namespace

extern "C" int fiery_bn_train_bwd(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                                  int C, const float* gamma, const float* mean, const float* invstd, int batch_stats, float* grad_in,
                                  int gi_ld, int C_store, float* dgamma, float* dbeta, float* workspace, fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && x && mean && invstd && grad_in && dgamma && dbeta && workspace, "bn_train_bwd: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && ld >= C && g_ld >= C && (!y || y_ld >= C) && C_store >= C && gi_ld >= C_store,
                  "bn_train_bwd: bad shape");
    FIERY_REQUIRE(C_store <= 4 * kBnThreads, "bn_train_bwd: at most %d stored channels", 4 * kBnThreads);
    const int rc = bn_bwd_reduce(grad_out, g_ld, x, ld, y, y_ld, n_pixels, C, mean, invstd, dgamma, dbeta, workspace, as_stream(stream));
    if (rc) return rc;
    return bn_bwd_dx(grad_out, g_ld, x, ld, y, y_ld, n_pixels, C, gamma, mean, invstd, dgamma, dbeta, batch_stats, n_pixels, grad_in, gi_ld,
                     C_store, as_stream(stream));
}

// ---- the same passes one at a time, for statistics that span several processes (SyncBatchNorm) ---------------------------------
extern "C" int fiery_bn_train_stats(const float* x, int ld, int64_t n_pixels, int C, float* mean, float* var, float* workspace,
                                    fiery_stream_t stream) {
    FIERY_REQUIRE(x && mean && var && workspace, "bn_train_stats: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && ld >= C && C <= 4 * kBnThreads, "bn_train_stats: bad shape");
    const BnShape s = bn_shape(n_pixels, C);
    const int Cp = (C + 3) / 4 * 4;
    hipLaunchKernelGGL(k_bn_partial_stats, dim3(s.blocks), dim3(kBnThreads), 0, as_stream(stream), x, ld, static_cast<long long>(n_pixels), C,
                       rows_vec(x, ld, C) ? 1 : 0, s, Cp, workspace);
    hipLaunchKernelGGL(k_bn_finish_stats, dim3(ceil_div(C, 4)), dim3(256), 0, as_stream(stream), x, workspace, s.blocks, Cp, C,
                       static_cast<long long>(n_pixels), 0.f, 0.f, mean, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                       static_cast<float*>(nullptr), var);
    return check_launch("bn_train_stats");
}

extern "C" int fiery_bn_apply(const float* x, int ld, int64_t n_pixels, int C, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, int relu, float* y, int y_ld, int C_store, fiery_stream_t stream) {
    FIERY_REQUIRE(x && y && mean && invstd, "bn_apply: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && ld >= C && C_store >= C && y_ld >= C_store, "bn_apply: bad shape");
    FIERY_REQUIRE(C_store <= 4 * kBnThreads, "bn_apply: at most %d channels", 4 * kBnThreads);
    const BnElemShape es = bn_elem_shape(n_pixels, C_store);
    const int y_vec = (C_store % 4 == 0 && y_ld % 4 == 0 && aligned16(y)) ? 1 : 0;
    hipLaunchKernelGGL(k_bn_apply, dim3(es.blocks), dim3(kBnThreads), 0, as_stream(stream), x, ld,
                       static_cast<long long>(n_pixels), C, rows_vec(x, ld, C) ? 1 : 0, mean, invstd, gamma, beta, relu, y, y_ld, C_store, y_vec,
                       es.groups, es.rows);
    return check_launch("bn_apply");
}

extern "C" int fiery_bn_train_bwd_sums(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                                       int C, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* workspace,
                                       fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && x && mean && invstd && dgamma && dbeta && workspace, "bn_train_bwd_sums: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && C > 0 && ld >= C && g_ld >= C && (!y || y_ld >= C) && C <= 4 * kBnThreads, "bn_train_bwd_sums: bad shape");
    return bn_bwd_reduce(grad_out, g_ld, x, ld, y, y_ld, n_pixels, C, mean, invstd, dgamma, dbeta, workspace, as_stream(stream));
}

extern "C" int fiery_bn_train_bwd_dx(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                                     int C, const float* gamma, const float* mean, const float* invstd, const float* dgamma,
                                     const float* dbeta, int64_t total_pixels, float* grad_in, int gi_ld, int C_store,
                                     fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && x && mean && invstd && dgamma && dbeta && grad_in, "bn_train_bwd_dx: null pointer");
    FIERY_REQUIRE(n_pixels > 0 && total_pixels >= n_pixels && C > 0 && ld >= C && g_ld >= C && (!y || y_ld >= C) && C_store >= C &&
                      gi_ld >= C_store, "bn_train_bwd_dx: bad shape");
    FIERY_REQUIRE(C_store <= 4 * kBnThreads, "bn_train_bwd_dx: at most %d stored channels", 4 * kBnThreads);
    return bn_bwd_dx(grad_out, g_ld, x, ld, y, y_ld, n_pixels, C, gamma, mean, invstd, dgamma, dbeta, 1, total_pixels, grad_in, gi_ld, C_store,
                     as_stream(stream));
}
