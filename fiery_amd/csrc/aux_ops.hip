// Small HBM-bound helpers of the BEV stack for gfx950: reductions, poolings, resampling, broadcasts and
// the layout changes at the API seams.  Each replaces a one-line ATen call of the reference (cited at
// the entry points in include/fiery_hip.h); all are unit-stride on the pixel-major (NHWC) layout.
#include "common.h"

#include <cstdlib>

#include <cstdint>

namespace fiery {
namespace {

constexpr int kMeanChunks = 64;

// stage 1: grid (kMeanChunks, n_img); 256 threads = 4 pixel lanes x 64 channel lanes
__global__ __launch_bounds__(256) void k_mean_partial(const float* __restrict__ in, int ld, long long outer_stride,
                                                      long long inner_stride, int n_inner, int HW, int C,
                                                      float* __restrict__ partial) {
    __shared__ float red[256];
    const int chunk = blockIdx.x, img = blockIdx.y;
    const int per = (HW + kMeanChunks - 1) / kMeanChunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const float* base = in + (img / n_inner) * outer_stride + (img % n_inner) * inner_stride;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cl;
        float s = 0.f;
        if (c < C)
            for (int p = p0 + pl; p < p1; p += 4) s += base[static_cast<long long>(p) * ld + c];
        red[threadIdx.x] = s;
        __syncthreads();
        if (pl == 0 && c < C)
            partial[(static_cast<long long>(img) * kMeanChunks + chunk) * C + c] =
                (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
        __syncthreads();
    }
}

__global__ void k_mean_final(const float* __restrict__ partial, int n_img, int C, int chunks, float inv_count,
                             float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img * C) return;
    const int img = i / C, c = i - img * C;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(static_cast<long long>(img) * chunks + k) * C + c];
    out[i] = s * inv_count;
}

// 16-byte variant of stage 1 for C % 4 == 0 and aligned rows: a thread owns four channels and every
// (256 / (C/4))-th pixel of its chunk, four independent accumulator quads per thread keep several loads in flight.
// grid (chunks, n_img), 256 threads; requires C/4 <= 256.
__global__ __launch_bounds__(256) void k_mean_partial4(const float* __restrict__ in, int ld, long long outer_stride,
                                                       long long inner_stride, int n_inner, int HW, int C, int chunks,
                                                       float* __restrict__ partial) {
    __shared__ float4 red[256];
    const int chunk = blockIdx.x, img = blockIdx.y;
    const int per = (HW + chunks - 1) / chunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const int c4n = C >> 2;                        // channel quads
    const int lanes = 256 / c4n;                   // pixel lanes
    const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const float* base = in + (img / n_inner) * outer_stride + (img % n_inner) * inner_stride + cq * 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (pl < lanes) {
        int p = p0 + pl;
        for (; p + 3 * lanes < p1; p += 4 * lanes) {
            const float4 a = *reinterpret_cast<const float4*>(base + static_cast<long long>(p) * ld);
            const float4 b = *reinterpret_cast<const float4*>(base + static_cast<long long>(p + lanes) * ld);
            const float4 c = *reinterpret_cast<const float4*>(base + static_cast<long long>(p + 2 * lanes) * ld);
            const float4 d = *reinterpret_cast<const float4*>(base + static_cast<long long>(p + 3 * lanes) * ld);
            s0.x += a.x;  s0.y += a.y;  s0.z += a.z;  s0.w += a.w;
            s1.x += b.x;  s1.y += b.y;  s1.z += b.z;  s1.w += b.w;
            s2.x += c.x;  s2.y += c.y;  s2.z += c.z;  s2.w += c.w;
            s3.x += d.x;  s3.y += d.y;  s3.z += d.z;  s3.w += d.w;
        }
        for (; p < p1; p += lanes) {
            const float4 a = *reinterpret_cast<const float4*>(base + static_cast<long long>(p) * ld);
            s0.x += a.x;  s0.y += a.y;  s0.z += a.z;  s0.w += a.w;
        }
    }
    red[threadIdx.x] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                                   (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (pl == 0) {
        float4 t = red[cq];
        for (int k = 1; k < lanes; ++k) {
            const float4 u = red[k * c4n + cq];
            t.x += u.x;  t.y += u.y;  t.z += u.z;  t.w += u.w;
        }
        *reinterpret_cast<float4*>(&partial[(static_cast<long long>(img) * chunks + chunk) * C + cq * 4]) = t;
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == FIERY_ACT_RELU) return fmaxf(v, 0.f);
    if (act == FIERY_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == FIERY_ACT_SWISH) return v * (1.0f / (1.0f + expf(-v)));
    return v;
}

__global__ void k_rowwise_dense(const float* __restrict__ v, int v_ld, int rows, int n_in, const float* __restrict__ W,
                                int w_ld, int w_col0, int n_out, float w_mul, const float* __restrict__ scale,
                                const float* __restrict__ shift, int act, int accumulate, float lo, float hi,
                                float* __restrict__ y, int y_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n_out) return;
    const int r = i / n_out, o = i - r * n_out;
    const float* wr = W + static_cast<long long>(o) * w_ld + w_col0;
    const float* vr = v + static_cast<long long>(r) * v_ld;
    float acc = 0.f;
    for (int j = 0; j < n_in; ++j) acc = fmaf(wr[j], vr[j], acc);
    float* dst = y + static_cast<long long>(r) * y_ld + o;
    acc *= w_mul;
    if (accumulate) acc += *dst;
    const float sc = scale ? scale[o] : 1.f, sh = shift ? shift[o] : 0.f;
    *dst = fminf(fmaxf(apply_act(fmaf(acc, sc, sh), act), lo), hi);
}

// n sequential fp32 additions of the constant v onto s, bit for bit, in O(number of binades crossed):
// while the running sum stays inside one binade (and keeps its sign) every step advances it by the same
// rounded increment, so after two real additions agree on that increment the rest of the binade is one
// multiply-add in double (exact: everything is a multiple of the binade's ulp).  Ties-to-even settle after one
// step, binade crossings and sign changes are walked with real additions.  Validated against the plain loop
// (tests/test_kernels_sim_aux.py) and against ATen's avg_pool3d.
__device__ float repeated_add(float s, float v, int n) {
    while (n > 0) {
        const float s1 = s + v;
        --n;
        if (n == 0 || s1 == s) return s1;            // done, or stagnated (|v| below half an ulp, or v == 0)
        const float s2 = s1 + v;
        --n;
        if (n == 0) return s2;
        const double d1 = static_cast<double>(s1) - static_cast<double>(s);
        const double d2 = static_cast<double>(s2) - static_cast<double>(s1);
        s = s2;
        if (d1 != d2 || fabsf(s1) < 1e-30f || fabsf(s2) < 1e-30f) continue;
        int e1, e2;
        frexpf(fabsf(s1), &e1);
        frexpf(fabsf(s2), &e2);
        if (e1 != e2 || (s1 > 0.f) != (s2 > 0.f)) continue;
        const double hi = ldexp(1.0, e2), lo = ldexp(0.5, e2), ulp = ldexp(1.0, e2 - 24);
        const double mag = fabs(static_cast<double>(s2));
        const double dm = s2 > 0.f ? d2 : -d2;        // change of magnitude per step
        const double room = dm > 0.0 ? (hi - ulp - mag) / dm : (mag - lo) / (-dm);
        long long k = static_cast<long long>(floor(room)) - 2;     // stay two steps clear of the binade edge
        if (k > 0) {
            if (k > n) k = n;
            s = static_cast<float>(static_cast<double>(s2) + static_cast<double>(k) * d2);
            n -= static_cast<int>(k);
        }
    }
    return s;
}

__global__ void k_sequential_window_mean(const float* __restrict__ prev, const float* __restrict__ cur, int rows, int n,
                                         int count_each, float* __restrict__ out, int out_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int r = i / n, j = i - r * n;
    float s = 0.f;
    int total = count_each;
    if (prev) {
        s = repeated_add(s, prev[static_cast<long long>(r) * n + j], count_each);
        total += count_each;
    }
    s = repeated_add(s, cur[static_cast<long long>(r) * n + j], count_each);
    out[static_cast<long long>(r) * out_ld + j] = s / static_cast<float>(total);
}

__global__ void k_latent_sample(const float* __restrict__ mu, const float* __restrict__ log_sigma,
                                const float* __restrict__ noise, int ld, int rows, int n, float* __restrict__ sample,
                                int sample_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int r = i / n, j = i - r * n;
    const float eps = noise ? noise[static_cast<long long>(r) * ld + j] : 0.f;
    sample[static_cast<long long>(r) * sample_ld + j] =
        mu[static_cast<long long>(r) * ld + j] + expf(log_sigma[static_cast<long long>(r) * ld + j]) * eps;
}

__global__ void k_maxpool2x2(const float* __restrict__ in, int in_ld, long long in_img_stride, int H, int W, int C, int Ho,
                             int Wo, float* __restrict__ out, int out_ld, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C);
    long long r = i / C;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float* base = in + static_cast<long long>(img) * in_img_stride;
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int y = 2 * yo + dy, x = 2 * xo + dx;
            // odd sizes were zero-padded by one row / column before pooling
            const float v = (y < H && x < W) ? base[(static_cast<long long>(y) * W + x) * in_ld + c] : 0.f;
            m = fmaxf(m, v);
        }
    out[((static_cast<long long>(img) * Ho + yo) * Wo + xo) * out_ld + c] = m;
}

// gradient of the above: the window's first maximum in row-major order takes the output's gradient
__global__ void k_maxpool2x2_bwd(const float* __restrict__ in, int in_ld, long long in_img_stride, const float* __restrict__ gy, int g_ld,
                                 int H, int W, int C, int Ho, int Wo, float* __restrict__ gx, int gi_ld, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C);
    long long r = i / C;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float* base = in + static_cast<long long>(img) * in_img_stride;
    float m = -INFINITY;
    int best = 0;
    for (int k = 0; k < 4; ++k) {
        const int y = 2 * yo + (k >> 1), x = 2 * xo + (k & 1);
        const float v = (y < H && x < W) ? base[(static_cast<long long>(y) * W + x) * in_ld + c] : 0.f;
        if (v > m || v != v) {                                  // ATen: (val > maxval) || isnan(val)
            m = v;
            best = k;
        }
    }
    const float g = gy[((static_cast<long long>(img) * Ho + yo) * Wo + xo) * g_ld + c];
    for (int k = 0; k < 4; ++k) {
        const int y = 2 * yo + (k >> 1), x = 2 * xo + (k & 1);
        if (y < H && x < W) gx[((static_cast<long long>(img) * H + y) * W + x) * gi_ld + c] = k == best ? g : 0.f;
    }
}

// bilinear x2, align_corners=False: source = dst / 2 - 0.25 clamped at 0, upper neighbour clamped at the edge
__global__ void k_upsample2x_add(const float* __restrict__ in, int in_ld, int H, int W, int C,
                                 const float* __restrict__ shift, const float* __restrict__ skip, int skip_ld,
                                 float* __restrict__ out, int out_ld, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C);
    long long r = i / C;
    const int Wo = 2 * W, Ho = 2 * H;
    const int x = static_cast<int>(r % Wo);
    r /= Wo;
    const int y = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float sy = fmaxf(0.5f * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (x + 0.5f) - 0.5f, 0.f);
    const int y0 = static_cast<int>(sy), x0 = static_cast<int>(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0;
    const float* base = in + static_cast<long long>(img) * H * W * in_ld + c;
    const float v00 = base[(static_cast<long long>(y0) * W + x0) * in_ld], v01 = base[(static_cast<long long>(y0) * W + x1) * in_ld];
    const float v10 = base[(static_cast<long long>(y1) * W + x0) * in_ld], v11 = base[(static_cast<long long>(y1) * W + x1) * in_ld];
    const float up = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    const long long po = (static_cast<long long>(img) * Ho + y) * Wo + x;
    out[po * out_ld + c] = up + (shift ? shift[c] : 0.f) + (skip ? skip[po * skip_ld + c] : 0.f);
}

// the same, four channels (16 bytes) per thread: C, the leading dimensions and the base addresses are multiples of 4
__global__ void k_upsample2x_add4(const float* __restrict__ in, int in_ld, int H, int W, int C4,
                                  const float* __restrict__ shift, const float* __restrict__ skip, int skip_ld,
                                  float* __restrict__ out, int out_ld, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C4) * 4;
    long long r = i / C4;
    const int Wo = 2 * W, Ho = 2 * H;
    const int x = static_cast<int>(r % Wo);
    r /= Wo;
    const int y = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float sy = fmaxf(0.5f * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (x + 0.5f) - 0.5f, 0.f);
    const int y0 = static_cast<int>(sy), x0 = static_cast<int>(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0;
    const float* base = in + static_cast<long long>(img) * H * W * in_ld + c;
    const float4 v00 = *reinterpret_cast<const float4*>(base + (static_cast<long long>(y0) * W + x0) * in_ld);
    const float4 v01 = *reinterpret_cast<const float4*>(base + (static_cast<long long>(y0) * W + x1) * in_ld);
    const float4 v10 = *reinterpret_cast<const float4*>(base + (static_cast<long long>(y1) * W + x0) * in_ld);
    const float4 v11 = *reinterpret_cast<const float4*>(base + (static_cast<long long>(y1) * W + x1) * in_ld);
    const long long po = (static_cast<long long>(img) * Ho + y) * Wo + x;
    const float4 sk = skip ? *reinterpret_cast<const float4*>(skip + po * skip_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = (1.f - ly) * ((1.f - lx) * v00.x + lx * v01.x) + ly * ((1.f - lx) * v10.x + lx * v11.x) + sh.x + sk.x;
    o.y = (1.f - ly) * ((1.f - lx) * v00.y + lx * v01.y) + ly * ((1.f - lx) * v10.y + lx * v11.y) + sh.y + sk.y;
    o.z = (1.f - ly) * ((1.f - lx) * v00.z + lx * v01.z) + ly * ((1.f - lx) * v10.z + lx * v11.z) + sh.z + sk.z;
    o.w = (1.f - ly) * ((1.f - lx) * v00.w + lx * v01.w) + ly * ((1.f - lx) * v10.w + lx * v11.w) + sh.w + sk.w;
    *reinterpret_cast<float4*>(out + po * out_ld + c) = o;
}

// Gradient of the x2 bilinear interpolation above with respect to its input (training; the transpose of the operator):
// input pixel (y, x) feeds output rows 2y-1, 2y, 2y+1, 2y+2 with weights 1/4, 3/4, 3/4, 1/4 - except at the borders, where
// the clamped neighbour folds onto the pixel itself (output row 0 takes input row 0 whole, output row 2H-1 takes input row
// H-1 whole) - and the same along x.  A gather: one thread = one input pixel x four channels, no atomics.
__global__ void k_upsample2x_bwd4(const float* __restrict__ g, int g_ld, int H, int W, int C4, float* __restrict__ gx, int gx_ld,
                                  long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C4) * 4;
    long long r = i / C4;
    const int x = static_cast<int>(r % W);
    r /= W;
    const int y = static_cast<int>(r % H);
    const int img = static_cast<int>(r / H);
    const int Wo = 2 * W;
    const float wy[4] = {y >= 1 ? 0.25f : 0.f, y == 0 ? 1.f : 0.75f, y == H - 1 ? 1.f : 0.75f, y <= H - 2 ? 0.25f : 0.f};
    const float wx[4] = {x >= 1 ? 0.25f : 0.f, x == 0 ? 1.f : 0.75f, x == W - 1 ? 1.f : 0.75f, x <= W - 2 ? 0.25f : 0.f};
    const float* base = g + static_cast<long long>(img) * 4 * H * W * g_ld + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        if (wy[dy] == 0.f) continue;
        const long long row = static_cast<long long>(2 * y - 1 + dy) * Wo;
        float4 line = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            if (wx[dx] == 0.f) continue;
            const float4 v = *reinterpret_cast<const float4*>(base + (row + 2 * x - 1 + dx) * g_ld);
            line.x += wx[dx] * v.x;  line.y += wx[dx] * v.y;  line.z += wx[dx] * v.z;  line.w += wx[dx] * v.w;
        }
        acc.x += wy[dy] * line.x;  acc.y += wy[dy] * line.y;  acc.z += wy[dy] * line.z;  acc.w += wy[dy] * line.w;
    }
    *reinterpret_cast<float4*>(gx + ((static_cast<long long>(img) * H + y) * W + x) * gx_ld + c) = acc;
}

// Weight gradient of the depthwise convolution (training of the image trunk):
//   dw[tap][c] += sum over images and output pixels of grad_out[img][y][x][c] * in[img][y s - pad_top + ky][x s - pad_left + kx][c]
// A workgroup = 16 channel quads x 16 row walkers: thread (q, r) keeps the K*K float4 partial sums of its four channels and
// walks output rows r, r + 16 gridDim.y ... of the (image, row) list; the sixteen walkers' sums of a tap meet in LDS and one
// lane per channel quad adds them to dw with fp32 atomics (K*K*C atomics per workgroup: the order of arrival decides the
// last bit, as for the dense weight gradient).  dw must be zero on entry.
template <int K>
__global__ __launch_bounds__(256) void k_depthwise_wgrad(const float* __restrict__ in, int in_ld, int H, int W, int C4,
                                                         const float* __restrict__ g, int g_ld, int Ho, int Wo, int stride, int pad_top,
                                                         int pad_left, long long n_rows, float* __restrict__ dw, int dw_ld) {
    __shared__ float4 part[16][16];
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 16 + q;
    const bool live = c4 < C4;
    const int c = c4 * 4;
    float4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && stride == 1 && (K == 3 || K == 5)) {
        // (round 5) stride 1: a thread walking along a row needs ONE new input column per step - the other K - 1 it read on the
        // steps before.  The K x K window lives in registers; the walk is unrolled K times so that the slot of column ix,
        // (ix + pad_left) mod K, is a compile-time index: 1 + K sixteen-byte loads per pixel instead of 1 + K * K (the
        // launches were load-issue bound: 320 us against ~95 us for their bytes).  Columns and rows outside the image are
        // zeros in the window - the sums see the same addends in the same order as the loop below.
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long row = static_cast<long long>(blockIdx.y) * 16 + r; row < n_rows; row += 16ll * gridDim.y) {
            const int img = static_cast<int>(row / Ho), y = static_cast<int>(row - static_cast<long long>(img) * Ho);
            const float* grow = g + (static_cast<long long>(img) * Ho + y) * Wo * g_ld + c;
            const float* ximg = in + static_cast<long long>(img) * H * W * in_ld + c;
            const float* xrow[K];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = y - pad_top + ky;
                xrow[ky] = (iy >= 0 && iy < H) ? ximg + static_cast<long long>(iy) * W * in_ld : nullptr;
            }
            float4 win[K][K];
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int j = 0; j < K - 1; ++j) {                                          // columns -pad_left .. K - 2 - pad_left
                    const int ix = j - pad_left;
                    win[ky][j] = (xrow[ky] && ix >= 0 && ix < W) ? *reinterpret_cast<const float4*>(xrow[ky] + static_cast<long long>(ix) * in_ld) : zero4;
                }
            for (int x0 = 0; x0 < Wo; x0 += K) {
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const int x = x0 + u;
                    if (x < Wo) {
                        const float4 gv = *reinterpret_cast<const float4*>(grow + static_cast<long long>(x) * g_ld);
                        const int ix_new = x + K - 1 - pad_left;
#pragma unroll
                        for (int ky = 0; ky < K; ++ky)
                            win[ky][(u + K - 1) % K] = (xrow[ky] && ix_new >= 0 && ix_new < W)
                                ? *reinterpret_cast<const float4*>(xrow[ky] + static_cast<long long>(ix_new) * in_ld) : zero4;
#pragma unroll
                        for (int ky = 0; ky < K; ++ky)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) {
                                const float4 xv = win[ky][(u + kx) % K];
                                float4& a = acc[ky * K + kx];
                                a.x = fmaf(gv.x, xv.x, a.x);  a.y = fmaf(gv.y, xv.y, a.y);  a.z = fmaf(gv.z, xv.z, a.z);  a.w = fmaf(gv.w, xv.w, a.w);
                            }
                    }
                }
            }
        }
    } else if (live) {
        for (long long row = static_cast<long long>(blockIdx.y) * 16 + r; row < n_rows; row += 16ll * gridDim.y) {
            const int img = static_cast<int>(row / Ho), y = static_cast<int>(row - static_cast<long long>(img) * Ho);
            const float* grow = g + (static_cast<long long>(img) * Ho + y) * Wo * g_ld + c;
            const float* ximg = in + static_cast<long long>(img) * H * W * in_ld + c;
            for (int x = 0; x < Wo; ++x) {
                const float4 gv = *reinterpret_cast<const float4*>(grow + static_cast<long long>(x) * g_ld);
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int iy = y * stride - pad_top + ky;
                    if (iy < 0 || iy >= H) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int ix = x * stride - pad_left + kx;
                        if (ix < 0 || ix >= W) continue;
                        const float4 xv = *reinterpret_cast<const float4*>(ximg + (static_cast<long long>(iy) * W + ix) * in_ld);
                        float4& a = acc[ky * K + kx];
                        a.x = fmaf(gv.x, xv.x, a.x);  a.y = fmaf(gv.y, xv.y, a.y);  a.z = fmaf(gv.z, xv.z, a.z);  a.w = fmaf(gv.w, xv.w, a.w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        part[r][q] = acc[t];
        __syncthreads();
        if (r == 0 && live) {
            float4 sum = part[0][q];
            for (int i = 1; i < 16; ++i) {
                const float4 v = part[i][q];
                sum.x += v.x;  sum.y += v.y;  sum.z += v.z;  sum.w += v.w;
            }
            float* d = dw + static_cast<long long>(t) * dw_ld + c;
            atomicAdd(d, sum.x);  atomicAdd(d + 1, sum.y);  atomicAdd(d + 2, sum.z);  atomicAdd(d + 3, sum.w);
        }
        __syncthreads();
    }
}

// Depthwise convolution (the image trunk's MBConv blocks), pixel-major: one thread = one output pixel x four channels.
// w is tap-major [k*k][C] so that a tap's four weights are one 16-byte load next to the four activations they meet.
// Zero padding is explicit (pad_top / pad_left before, whatever Hout / Wout imply after): the trunk's "static same"
// padding is asymmetric.  Folded BatchNorm and the activation are applied on the way out.
__global__ void k_depthwise4(const float* __restrict__ in, int in_ld, int H, int W, int C4, const float* __restrict__ w,
                             int w_ld, int k, int stride, int pad_top, int pad_left, int Ho, int Wo,
                             const float* __restrict__ scale, const float* __restrict__ shift, int act,
                             float* __restrict__ out, int out_ld, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C4) * 4;
    long long r = i / C4;
    const int x = static_cast<int>(r % Wo);
    r /= Wo;
    const int y = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float* base = in + static_cast<long long>(img) * H * W * in_ld + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < k; ++ky) {
        const int iy = y * stride - pad_top + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = x * stride - pad_left + kx;
            if (ix < 0 || ix >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(base + (static_cast<long long>(iy) * W + ix) * in_ld);
            const float4 t = *reinterpret_cast<const float4*>(w + static_cast<long long>(ky * k + kx) * w_ld + c);
            acc.x = fmaf(v.x, t.x, acc.x);  acc.y = fmaf(v.y, t.y, acc.y);  acc.z = fmaf(v.z, t.z, acc.z);  acc.w = fmaf(v.w, t.w, acc.w);
        }
    }
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o;
    o.x = apply_act(fmaf(acc.x, sc.x, sh.x), act);  o.y = apply_act(fmaf(acc.y, sc.y, sh.y), act);
    o.z = apply_act(fmaf(acc.z, sc.z, sh.z), act);  o.w = apply_act(fmaf(acc.w, sc.w, sh.w), act);
    *reinterpret_cast<float4*>(out + ((static_cast<long long>(img) * Ho + y) * Wo + x) * out_ld + c) = o;
}

// The trunk's four depthwise shapes (K = 3 / 5, stride 1 / 2) with a register tile: one thread = four neighbouring
// output pixels of a row x four channels.  A row of the window is loaded once - (4 - 1) * S + K columns - and feeds all
// four outputs, which takes 18 / 40 / 27 / 55 16-byte loads per thread where the one-pixel form needs 36 / 100 / 36 / 100
// for the same outputs.
template <int K, int S>
__global__ void k_depthwise4_tiled(const float* __restrict__ in, int in_ld, int H, int W, int C4, const float* __restrict__ w,
                                   int w_ld, int pad_top, int pad_left, int Ho, int Wo, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int act, float* __restrict__ out, int out_ld, long long total) {
    constexpr int TX = 4, NCOL = (TX - 1) * S + K;
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C4) * 4;
    long long r = i / C4;
    const int Wt = (Wo + TX - 1) / TX;
    const int x0 = static_cast<int>(r % Wt) * TX;
    r /= Wt;
    const int y = static_cast<int>(r % Ho);
    const int img = static_cast<int>(r / Ho);
    const float* base = in + static_cast<long long>(img) * H * W * in_ld + c;
    float4 acc[TX];
#pragma unroll
    for (int t = 0; t < TX; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ix0 = x0 * S - pad_left;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = y * S - pad_top + ky;
        if (iy < 0 || iy >= H) continue;
        float4 col[NCOL], wt[K];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int ix = ix0 + j;
            col[j] = (ix >= 0 && ix < W) ? *reinterpret_cast<const float4*>(base + (static_cast<long long>(iy) * W + ix) * in_ld)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wt[kx] = *reinterpret_cast<const float4*>(w + static_cast<long long>(ky * K + kx) * w_ld + c);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const float4 v = col[t * S + kx];
                acc[t].x = fmaf(v.x, wt[kx].x, acc[t].x);  acc[t].y = fmaf(v.y, wt[kx].y, acc[t].y);
                acc[t].z = fmaf(v.z, wt[kx].z, acc[t].z);  acc[t].w = fmaf(v.w, wt[kx].w, acc[t].w);
            }
    }
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < TX; ++t) {
        if (x0 + t >= Wo) break;
        float4 o;
        o.x = apply_act(fmaf(acc[t].x, sc.x, sh.x), act);  o.y = apply_act(fmaf(acc[t].y, sc.y, sh.y), act);
        o.z = apply_act(fmaf(acc[t].z, sc.z, sh.z), act);  o.w = apply_act(fmaf(acc[t].w, sc.w, sh.w), act);
        *reinterpret_cast<float4*>(out + ((static_cast<long long>(img) * Ho + y) * Wo + x0 + t) * out_ld + c) = o;
    }
}

// x[img][pixel][c] *= gate[img][c]   (squeeze-and-excite), four channels per thread, in place
__global__ void k_scale_channels4(float* __restrict__ x, int ld, int HW, int C4, const float* __restrict__ gate, int gate_ld,
                                  long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C4) * 4;
    const long long px = i / C4;
    const int img = static_cast<int>(px / HW);
    float4* p = reinterpret_cast<float4*>(x + px * ld + c);
    const float4 g = *reinterpret_cast<const float4*>(gate + static_cast<long long>(img) * gate_ld + c);
    float4 v = *p;
    v.x *= g.x;  v.y *= g.y;  v.z *= g.z;  v.w *= g.w;
    *p = v;
}

// ------------------------------------------------------------------------------------------------
// the step after the path: per-frame instance segmentation (evaluation)
// ------------------------------------------------------------------------------------------------
// One workgroup per frame; replaces the ATen chain of fiery/utils/instance.py:80-144 (threshold, 3x3 max-pool NMS,
// nonzero, distance matrix + argmin over <= max_centers centres, foreground mask, unique + renumbering):
//   1. centres = pixels above the threshold that equal the maximum of their 3x3 neighbourhood (of thresholded values,
//      -inf outside the image), collected in row-major order by ballot/prefix compaction - `torch.nonzero`'s order -,
//      the first max_centers kept;
//   2. every pixel joins the first centre at minimal distance from (pixel + offset); background pixels get 0;
//   3. the ids that occur are renumbered 0, 1, 2, ... in ascending order (`torch.unique` + `update_instance_ids`).
// Index arithmetic throughout; the only floating-point decision is the argmin, evaluated as sqrtf(dx*dx + dy*dy) with
// separately rounded products (ATen's vectorised norm differs from that in the last bit of some distances, which can
// only matter for a pixel whose two nearest centres are equidistant to within one ulp).
constexpr int kMaxInstanceCenters = 256;
#pragma clang fp contract(off)
__global__ __launch_bounds__(1024) void k_instance_segmentation(const float* __restrict__ center, const float* __restrict__ offset,
                                                                const unsigned char* __restrict__ foreground, int H, int W,
                                                                float threshold, int max_centers, int* __restrict__ seg,
                                                                int* __restrict__ centers, int* __restrict__ n_centers) {
    __shared__ int c_y[kMaxInstanceCenters], c_x[kMaxInstanceCenters];
    __shared__ int present[kMaxInstanceCenters + 1], remap[kMaxInstanceCenters + 1];
    __shared__ int wave_count[16];
    __shared__ int running;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int HW = H * W;
    const float* c = center + static_cast<long long>(f) * HW;
    const float* off = offset + static_cast<long long>(f) * 2 * HW;
    const unsigned char* fg = foreground + static_cast<long long>(f) * HW;
    int* out = seg + static_cast<long long>(f) * HW;
    if (tid == 0) running = 0;
    for (int k = tid; k <= kMaxInstanceCenters; k += blockDim.x) present[k] = 0;
    __syncthreads();
    auto thresholded = [&](int y, int x) {
        const float v = c[y * W + x];
        return v > threshold ? v : -1.0f;
    };
    // 1. ordered compaction of the local maxima
    for (int p0 = 0; p0 < HW; p0 += blockDim.x) {
        const int p = p0 + tid;
        bool keep = false;
        if (p < HW) {
            const int y = p / W, x = p - y * W;
            const float v = thresholded(y, x);
            float m = v;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, thresholded(yy, xx));
                }
            keep = v == m && v > 0.f;
        }
        const unsigned long long ballot = __ballot(keep);
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        __syncthreads();
        int slot = running + before;
        for (int k = 0; k < wave; ++k) slot += wave_count[k];
        if (keep && slot < max_centers) {
            c_y[slot] = p / W;
            c_x[slot] = p % W;
        }
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int k = 0; k < n_waves; ++k) total += wave_count[k];
            running += total;
        }
        __syncthreads();
    }
    const int n = min(running, max_centers);
    if (tid == 0) n_centers[f] = n;
    for (int k = tid; k < max_centers; k += blockDim.x) {
        centers[(static_cast<long long>(f) * max_centers + k) * 2] = k < n ? c_y[k] : -1;
        centers[(static_cast<long long>(f) * max_centers + k) * 2 + 1] = k < n ? c_x[k] : -1;
    }
    if (n == 0) {                                       // instance.py:129-131: nothing detected, all background
        for (int p = tid; p < HW; p += blockDim.x) out[p] = 0;
        return;
    }
    // 2. nearest centre of (pixel + offset)
    for (int p = tid; p < HW; p += blockDim.x) {
        const int y = p / W, x = p - y * W;
        const float ly = static_cast<float>(y) + off[p], lx = static_cast<float>(x) + off[HW + p];
        float best = 0.f;
        int arg = 0;
        for (int k = 0; k < n; ++k) {
            const float dy = static_cast<float>(c_y[k]) - ly, dx = static_cast<float>(c_x[k]) - lx;
            const float d = sqrtf(dy * dy + dx * dx);
            if (k == 0 || d < best) {
                best = d;
                arg = k;
            }
        }
        const int id = fg[p] ? arg + 1 : 0;
        out[p] = id;
        present[id] = 1;
    }
    __syncthreads();
    // 3. renumber the ids that occur
    if (tid == 0) {
        int next = 0;
        for (int k = 0; k <= n; ++k) {
            remap[k] = next;
            next += present[k];
        }
    }
    __syncthreads();
    for (int p = tid; p < HW; p += blockDim.x) out[p] = remap[out[p]];
}
#pragma clang fp contract(fast)

// Squeeze-and-excite gate of one image per workgroup: gate = sigmoid(W2 . swish(W1 . mean + b1) + b2).
// The channel means sit in LDS; a thread owns channels t, t + 256, ... and forms its share of every hidden unit (W1 rows
// are read unit-stride across the workgroup), the shares meet in LDS and are summed in thread order (reproducible),
// then a thread per channel finishes the gate.  C <= 1024, hidden units <= 64.
constexpr int kSeMaxHidden = 64;
// `mean` holds `chunks` rows of partial sums per image (row stride ld) that add up to the channel sums; inv_count turns
// them into means (chunks = 1, inv_count = 1: the means themselves).
__global__ __launch_bounds__(256) void k_se_gate(const float* __restrict__ mean, int ld, int chunks, float inv_count, int C,
                                                 const float* __restrict__ w1, const float* __restrict__ b1, int sq,
                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                 float* __restrict__ gate, int gate_ld) {
    // (round 4: the first dense layer used to keep 64 partial sums per thread and have `sq` threads add up 256 of them
    // each from LDS - 41 us per call, 0.9 ms per pass of the trunk; now a wavefront owns a hidden unit at a time: its lanes
    // stride over the channels and meet in a shuffle tree, every sum in a fixed order)
    __shared__ float means[1024];
    __shared__ float hidden[kSeMaxHidden];
    const int img = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* m = mean + static_cast<long long>(img) * chunks * ld;
    for (int c = t; c < C; c += 256) {
        float mv = 0.f;
        for (int k0 = 0; k0 < chunks; k0 += 8) {               // eight partial rows in flight, summed in order
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = k0 + j < chunks ? m[static_cast<long long>(k0 + j) * ld + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) mv += pv[j];
        }
        means[c] = mv * inv_count;
    }
    __syncthreads();
    for (int s_ = wave; s_ < sq; s_ += 4) {
        const float* wr = w1 + static_cast<long long>(s_) * C;
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(wr[c], means[c], a);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
        if (lane == 0) hidden[s_] = apply_act(a + b1[s_], FIERY_ACT_SWISH);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = b2[c];
        const float* wr = w2 + static_cast<long long>(c) * sq;
        for (int s0 = 0; s0 < sq; s0 += 8) {                  // eight weights in flight, summed in order
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = s0 + j < sq ? wr[s0 + j] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < sq) a = fmaf(wv[j], hidden[s0 + j], a);
        }
        gate[static_cast<long long>(img) * gate_ld + c] = apply_act(a, FIERY_ACT_SIGMOID);
    }
}

__global__ void k_broadcast(const float* __restrict__ v, int v_ld, int HW, int C, float* __restrict__ out, int out_ld,
                            long long out_img_stride, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = static_cast<int>(i % C);
    long long r = i / C;
    const int p = static_cast<int>(r % HW);
    const int img = static_cast<int>(r / HW);
    out[img * out_img_stride + static_cast<long long>(p) * out_ld + c] = v[static_cast<long long>(img) * v_ld + c];
}

// 64-pixel tiles through LDS so both sides are unit-stride
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, int C, int HW, float* __restrict__ out,
                                                      int out_ld, long long out_img_stride) {
    HIP_DYNAMIC_SHARED(float, tile)   // [64][C + 1]
    const int img = blockIdx.y, p0 = blockIdx.x * 64;
    const int npx = min(64, HW - p0);
    const int row = C + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = in + static_cast<long long>(img) * C * HW + p0;
    for (int c = wave; c < C; c += 4)
        if (lane < npx) tile[lane * row + c] = src[static_cast<long long>(c) * HW + lane];
    __syncthreads();
    float* dst = out + img * out_img_stride + static_cast<long long>(p0) * out_ld;
    for (int i = threadIdx.x; i < npx * C; i += blockDim.x) {
        const int px = i / C, c = i - px * C;
        dst[static_cast<long long>(px) * out_ld + c] = tile[px * row + c];
    }
}

__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, int in_ld, long long in_img_stride, int C,
                                                      int HW, float* __restrict__ out) {
    HIP_DYNAMIC_SHARED(float, tile)   // [64][C + 1]
    const int img = blockIdx.y, p0 = blockIdx.x * 64;
    const int npx = min(64, HW - p0);
    const int row = C + 1;
    const float* src = in + img * in_img_stride + static_cast<long long>(p0) * in_ld;
    for (int i = threadIdx.x; i < npx * C; i += blockDim.x) {
        const int px = i / C, c = i - px * C;
        tile[px * row + c] = src[static_cast<long long>(px) * in_ld + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* dst = out + static_cast<long long>(img) * C * HW + p0;
    for (int c = wave; c < C; c += 4)
        if (lane < npx) dst[static_cast<long long>(c) * HW + lane] = tile[lane * row + c];
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_spatial_mean(const float* in, int in_ld, int64_t outer_stride, int n_outer, int64_t inner_stride,
                                  int n_inner, int n_pixels, int C, float* out, float* workspace, fiery_stream_t stream) {
    FIERY_REQUIRE(in && out && workspace, "spatial_mean: null pointer");
    FIERY_REQUIRE(n_outer > 0 && n_inner > 0 && n_pixels > 0 && C > 0 && in_ld >= C, "spatial_mean: bad shape");
    const int n_img = n_outer * n_inner;
    // 16-byte rows when the layout allows; the chunk count stays kMeanChunks (the workspace contract)
    const bool vec4 = C % 4 == 0 && C / 4 <= 256 && in_ld % 4 == 0 && outer_stride % 4 == 0 && inner_stride % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL(k_mean_partial4, dim3(kMeanChunks, n_img), dim3(256), 0, as_stream(stream), in, in_ld,
                           static_cast<long long>(outer_stride), static_cast<long long>(inner_stride), n_inner, n_pixels, C,
                           kMeanChunks, workspace);
    else
        hipLaunchKernelGGL(k_mean_partial, dim3(kMeanChunks, n_img), dim3(256), 0, as_stream(stream), in, in_ld,
                           static_cast<long long>(outer_stride), static_cast<long long>(inner_stride), n_inner, n_pixels, C,
                           workspace);
    int rc = check_launch("spatial_mean(partial)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_mean_final, dim3(ceil_div(n_img * C, 256)), dim3(256), 0, as_stream(stream), workspace, n_img, C,
                       kMeanChunks, 1.0f / static_cast<float>(n_pixels), out);
    return check_launch("spatial_mean(final)");
}

extern "C" int fiery_rowwise_dense(const float* v, int v_ld, int rows, int n_in, const float* W, int w_ld, int w_col0,
                                   int n_out, float w_mul, const float* scale, const float* shift, int act, int accumulate,
                                   float lo, float hi, float* y, int y_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(v && W && y && rows > 0 && n_in > 0 && n_out > 0, "rowwise_dense: bad argument");
    FIERY_REQUIRE(w_col0 >= 0 && w_col0 + n_in <= w_ld && v_ld >= n_in && y_ld >= n_out, "rowwise_dense: bad strides");
    hipLaunchKernelGGL(k_rowwise_dense, dim3(ceil_div(rows * n_out, 128)), dim3(128), 0, as_stream(stream), v, v_ld, rows,
                       n_in, W, w_ld, w_col0, n_out, w_mul, scale, shift, act, accumulate, lo, hi, y, y_ld);
    return check_launch("rowwise_dense");
}

extern "C" int fiery_sequential_window_mean(const float* prev, const float* cur, int rows, int n, int count_each, float* out,
                                            int out_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(cur && out && rows > 0 && n > 0 && count_each > 0 && out_ld >= n, "sequential_window_mean: bad argument");
    hipLaunchKernelGGL(k_sequential_window_mean, dim3(ceil_div(rows * n, 64)), dim3(64), 0, as_stream(stream), prev, cur,
                       rows, n, count_each, out, out_ld);
    return check_launch("sequential_window_mean");
}

extern "C" int fiery_latent_sample(const float* mu, const float* log_sigma, const float* noise, int ld, int rows, int n,
                                   float* sample, int sample_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(mu && log_sigma && sample && rows > 0 && n > 0 && ld >= n && sample_ld >= n, "latent_sample: bad argument");
    hipLaunchKernelGGL(k_latent_sample, dim3(ceil_div(rows * n, 128)), dim3(128), 0, as_stream(stream), mu, log_sigma, noise,
                       ld, rows, n, sample, sample_ld);
    return check_launch("latent_sample");
}

extern "C" int fiery_maxpool2x2_nhwc(const float* in, int in_ld, int64_t in_img_stride, int n_img, int H, int W, int C,
                                     float* out, int out_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(in && out && n_img > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad argument");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = static_cast<long long>(n_img) * Ho * Wo * C;
    const long long istride = in_img_stride > 0 ? in_img_stride : static_cast<long long>(H) * W * in_ld;
    hipLaunchKernelGGL(k_maxpool2x2, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), in, in_ld, istride, H, W, C, Ho,
                       Wo, out, out_ld, total);
    return check_launch("maxpool2x2");
}

extern "C" int fiery_maxpool2x2_bwd_nhwc(const float* in, int in_ld, int64_t in_img_stride, const float* grad_out, int g_ld, int n_img,
                                         int H, int W, int C, float* grad_in, int gi_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(in && grad_out && grad_in && n_img > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_bwd: bad argument");
    FIERY_REQUIRE(in_ld >= C && g_ld >= C && gi_ld >= C, "maxpool2x2_bwd: leading dimension below the channel count");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = static_cast<long long>(n_img) * Ho * Wo * C;
    const long long istride = in_img_stride > 0 ? in_img_stride : static_cast<long long>(H) * W * in_ld;
    hipLaunchKernelGGL(k_maxpool2x2_bwd, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), in, in_ld, istride, grad_out, g_ld,
                       H, W, C, Ho, Wo, grad_in, gi_ld, total);
    return check_launch("maxpool2x2_bwd");
}

extern "C" int fiery_upsample2x_add_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C, const float* shift,
                                         const float* skip, int skip_ld, float* out, int out_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(in && out && n_img > 0 && H > 0 && W > 0 && C > 0, "upsample2x_add: bad argument");
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (C % 4 == 0 && in_ld % 4 == 0 && (!skip || skip_ld % 4 == 0) && out_ld % 4 == 0 && a16(in) && (!skip || a16(skip)) && a16(out) &&
        (!shift || a16(shift))) {
        const long long total4 = static_cast<long long>(n_img) * 4 * H * W * (C / 4);
        hipLaunchKernelGGL(k_upsample2x_add4, dim3(ceil_div(total4, 256)), dim3(256), 0, as_stream(stream), in, in_ld, H, W,
                           C / 4, shift, skip, skip_ld, out, out_ld, total4);
        return check_launch("upsample2x_add");
    }
    const long long total = static_cast<long long>(n_img) * 4 * H * W * C;
    hipLaunchKernelGGL(k_upsample2x_add, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), in, in_ld, H, W, C,
                       shift, skip, skip_ld, out, out_ld, total);
    return check_launch("upsample2x_add");
}

extern "C" int fiery_upsample2x_bwd_nhwc(const float* grad_out, int g_ld, int n_img, int H, int W, int C, float* grad_in, int gi_ld,
                                         fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && grad_in && n_img > 0 && H > 0 && W > 0 && C > 0, "upsample2x_bwd: bad argument");
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    FIERY_REQUIRE(C % 4 == 0 && g_ld % 4 == 0 && gi_ld % 4 == 0 && g_ld >= C && gi_ld >= C && a16(grad_out) && a16(grad_in),
                  "upsample2x_bwd: channels and leading dimensions must be multiples of 4, buffers 16-byte aligned");
    const long long total4 = static_cast<long long>(n_img) * H * W * (C / 4);
    hipLaunchKernelGGL(k_upsample2x_bwd4, dim3(ceil_div(total4, 256)), dim3(256), 0, as_stream(stream), grad_out, g_ld, H, W, C / 4,
                       grad_in, gi_ld, total4);
    return check_launch("upsample2x_bwd");
}

extern "C" int fiery_depthwise_conv_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C, const float* w, int w_ld,
                                         int k, int stride, int pad_top, int pad_left, int Hout, int Wout, const float* scale,
                                         const float* shift, int act, float* out, int out_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(in && w && out && n_img > 0 && H > 0 && W > 0 && C > 0, "depthwise_conv: bad argument");
    FIERY_REQUIRE(k > 0 && k <= 7 && (stride == 1 || stride == 2) && pad_top >= 0 && pad_left >= 0 && Hout > 0 && Wout > 0,
                  "depthwise_conv: bad kernel geometry");
    // (a window may lie wholly in the padding - its output is the shift alone: the input gradient, computed with this same
    // kernel on the zero-stuffed output gradient, has such rows when no window of the forward pass reached them)
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    FIERY_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && w_ld % 4 == 0 && in_ld >= C && out_ld >= C && w_ld >= C &&
                  a16(in) && a16(out) && a16(w) && (!scale || a16(scale)) && (!shift || a16(shift)),
                  "depthwise_conv: channels and leading dimensions must be multiples of 4, pointers 16-byte aligned");
    FIERY_REQUIRE(act >= FIERY_ACT_NONE && act <= FIERY_ACT_SWISH, "depthwise_conv: unknown activation");
    bool tiled = k == 3 || k == 5;                     // the register-tiled kernels; other sizes take the general one
    if (const char* forced = getenv("FIERY_DEPTHWISE_TILED")) tiled = tiled && atoi(forced) != 0;          // tuning / tests
    if (tiled) {
        const long long total_t = static_cast<long long>(n_img) * Hout * ((Wout + 3) / 4) * (C / 4);
#define FIERY_DW_LAUNCH(K_, S_)                                                                                            \
    hipLaunchKernelGGL((k_depthwise4_tiled<K_, S_>), dim3(ceil_div(total_t, 256)), dim3(256), 0, as_stream(stream), in, in_ld, H, \
                       W, C / 4, w, w_ld, pad_top, pad_left, Hout, Wout, scale, shift, act, out, out_ld, total_t)
        if (k == 3 && stride == 1) FIERY_DW_LAUNCH(3, 1);
        else if (k == 3) FIERY_DW_LAUNCH(3, 2);
        else if (stride == 1) FIERY_DW_LAUNCH(5, 1);
        else FIERY_DW_LAUNCH(5, 2);
#undef FIERY_DW_LAUNCH
        return check_launch("depthwise_conv");
    }
    const long long total = static_cast<long long>(n_img) * Hout * Wout * (C / 4);
    hipLaunchKernelGGL(k_depthwise4, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), in, in_ld, H, W, C / 4, w, w_ld,
                       k, stride, pad_top, pad_left, Hout, Wout, scale, shift, act, out, out_ld, total);
    return check_launch("depthwise_conv");
}

extern "C" int fiery_depthwise_conv_wgrad_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C, const float* grad_out,
                                               int g_ld, int Hout, int Wout, int k, int stride, int pad_top, int pad_left, float* dw,
                                               int dw_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(in && grad_out && dw && n_img > 0 && H > 0 && W > 0 && C > 0 && Hout > 0 && Wout > 0, "depthwise_conv_wgrad: bad argument");
    FIERY_REQUIRE((k == 1 || k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2) && pad_top >= 0 && pad_left >= 0,
                  "depthwise_conv_wgrad: kernel sizes 1, 3, 5, 7 and strides 1, 2");
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    FIERY_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && g_ld % 4 == 0 && in_ld >= C && g_ld >= C && dw_ld >= C && a16(in) && a16(grad_out),
                  "depthwise_conv_wgrad: channels and leading dimensions must be multiples of 4, pointers 16-byte aligned");
    const long long n_rows = static_cast<long long>(n_img) * Hout;
    const int c_groups = ceil_div(C / 4, 16);
    long long walkers = ceil_div(n_rows, 16);                            // workgroups along the rows: fill the chip, a few rows each
    const long long want = 2048 / c_groups > 0 ? 2048 / c_groups : 1;
    if (walkers > want) walkers = want;
    const dim3 grid(static_cast<unsigned>(c_groups), static_cast<unsigned>(walkers));
#define FIERY_DWW_LAUNCH(K_)                                                                                              \
    hipLaunchKernelGGL((k_depthwise_wgrad<K_>), grid, dim3(256), 0, as_stream(stream), in, in_ld, H, W, C / 4, grad_out, g_ld, Hout, \
                       Wout, stride, pad_top, pad_left, n_rows, dw, dw_ld)
    if (k == 1) FIERY_DWW_LAUNCH(1);
    else if (k == 3) FIERY_DWW_LAUNCH(3);
    else if (k == 5) FIERY_DWW_LAUNCH(5);
    else FIERY_DWW_LAUNCH(7);
#undef FIERY_DWW_LAUNCH
    return check_launch("depthwise_conv_wgrad");
}

extern "C" int fiery_instance_segmentation(const float* center, const float* offset, const uint8_t* foreground, int n_frames,
                                           int H, int W, float conf_threshold, int max_centers, int32_t* instance_seg,
                                           int32_t* centers, int32_t* n_centers, fiery_stream_t stream) {
    FIERY_REQUIRE(center && offset && foreground && instance_seg && centers && n_centers, "instance_segmentation: null pointer");
    FIERY_REQUIRE(n_frames > 0 && H > 0 && W > 0 && static_cast<long long>(H) * W < (1ll << 30), "instance_segmentation: bad shape");
    FIERY_REQUIRE(max_centers > 0 && max_centers <= kMaxInstanceCenters, "instance_segmentation: at most %d centres per frame",
                  kMaxInstanceCenters);
    hipLaunchKernelGGL(k_instance_segmentation, dim3(n_frames), dim3(1024), 0, as_stream(stream), center, offset, foreground, H, W,
                       conf_threshold, max_centers, instance_seg, centers, n_centers);
    return check_launch("instance_segmentation");
}

extern "C" int fiery_se_gate(const float* mean, int mean_ld, int n_img, int C, const float* w1, const float* b1, int hidden,
                             const float* w2, const float* b2, float* gate, int gate_ld, fiery_stream_t stream) {
    FIERY_REQUIRE(mean && w1 && b1 && w2 && b2 && gate && n_img > 0, "se_gate: bad argument");
    FIERY_REQUIRE(C > 0 && C <= 1024 && hidden > 0 && hidden <= kSeMaxHidden && mean_ld >= C && gate_ld >= C,
                  "se_gate: at most 1024 channels and %d hidden units", kSeMaxHidden);
    hipLaunchKernelGGL(k_se_gate, dim3(n_img), dim3(256), 0, as_stream(stream), mean, mean_ld, 1, 1.0f, C, w1, b1, hidden, w2, b2,
                       gate, gate_ld);
    return check_launch("se_gate");
}

extern "C" int fiery_se_gate_nhwc(const float* x, int ld, int64_t img_stride, int n_img, int n_pixels, int C, const float* w1,
                                  const float* b1, int hidden, const float* w2, const float* b2, float* gate, int gate_ld,
                                  float* workspace, fiery_stream_t stream) {
    FIERY_REQUIRE(x && w1 && b1 && w2 && b2 && gate && workspace && n_img > 0 && n_pixels > 0, "se_gate_nhwc: bad argument");
    FIERY_REQUIRE(C > 0 && C <= 1024 && hidden > 0 && hidden <= kSeMaxHidden && ld >= C && gate_ld >= C,
                  "se_gate_nhwc: at most 1024 channels and %d hidden units", kSeMaxHidden);
    const bool vec = C % 4 == 0 && ld % 4 == 0 && img_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    // partial rows per image: enough workgroups to fill the chip (chunks * n_img >= ~512), few enough that the gate
    // kernel - one workgroup per image, latency-bound - has little to add up
    int chunks = kMeanChunks;
    if (vec) {
        chunks = 512 / n_img;
        chunks = chunks < 4 ? 4 : (chunks > kMeanChunks ? kMeanChunks : chunks);
    }
    if (vec)
        hipLaunchKernelGGL(k_mean_partial4, dim3(chunks, n_img), dim3(256), 0, as_stream(stream), x, ld,
                           static_cast<long long>(img_stride), 0ll, 1, n_pixels, C, chunks, workspace);
    else
        hipLaunchKernelGGL(k_mean_partial, dim3(kMeanChunks, n_img), dim3(256), 0, as_stream(stream), x, ld,
                           static_cast<long long>(img_stride), 0ll, 1, n_pixels, C, workspace);
    const int rc = check_launch("se_gate_nhwc (channel sums)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_se_gate, dim3(n_img), dim3(256), 0, as_stream(stream), workspace, C, chunks,
                       1.0f / static_cast<float>(n_pixels), C, w1, b1, hidden, w2, b2, gate, gate_ld);
    return check_launch("se_gate_nhwc");
}

extern "C" int fiery_scale_channels_nhwc(float* x, int ld, int n_img, int HW, int C, const float* gate, int gate_ld,
                                         fiery_stream_t stream) {
    FIERY_REQUIRE(x && gate && n_img > 0 && HW > 0 && C > 0, "scale_channels: bad argument");
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    FIERY_REQUIRE(C % 4 == 0 && ld % 4 == 0 && gate_ld % 4 == 0 && ld >= C && gate_ld >= C && a16(x) && a16(gate),
                  "scale_channels: channels and leading dimensions must be multiples of 4, pointers 16-byte aligned");
    const long long total = static_cast<long long>(n_img) * HW * (C / 4);
    hipLaunchKernelGGL(k_scale_channels4, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), x, ld, HW, C / 4, gate,
                       gate_ld, total);
    return check_launch("scale_channels");
}

extern "C" int fiery_broadcast_nhwc(const float* v, int v_ld, int n_img, int HW, int C, float* out, int out_ld,
                                    int64_t out_img_stride, fiery_stream_t stream) {
    FIERY_REQUIRE(v && out && n_img > 0 && HW > 0 && C > 0, "broadcast: bad argument");
    const long long total = static_cast<long long>(n_img) * HW * C;
    hipLaunchKernelGGL(k_broadcast, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), v, v_ld, HW, C, out, out_ld,
                       static_cast<long long>(out_img_stride), total);
    return check_launch("broadcast");
}

extern "C" int fiery_nchw_to_nhwc(const float* in, int n_img, int C, int HW, float* out, int out_ld, int64_t out_img_stride,
                                  fiery_stream_t stream) {
    FIERY_REQUIRE(in && out && n_img > 0 && C > 0 && HW > 0 && out_ld >= C, "nchw_to_nhwc: bad argument");
    FIERY_REQUIRE(static_cast<size_t>(64) * (C + 1) * sizeof(float) <= 160 * 1024, "nchw_to_nhwc: too many channels");
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(ceil_div(HW, 64), n_img), dim3(256), static_cast<size_t>(64) * (C + 1) * sizeof(float),
                       as_stream(stream), in, C, HW, out, out_ld, static_cast<long long>(out_img_stride));
    return check_launch("nchw_to_nhwc");
}

extern "C" int fiery_nhwc_to_nchw(const float* in, int in_ld, int64_t in_img_stride, int n_img, int C, int HW, float* out,
                                  fiery_stream_t stream) {
    FIERY_REQUIRE(in && out && n_img > 0 && C > 0 && HW > 0 && in_ld >= C, "nhwc_to_nchw: bad argument");
    FIERY_REQUIRE(static_cast<size_t>(64) * (C + 1) * sizeof(float) <= 160 * 1024, "nhwc_to_nchw: too many channels");
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(ceil_div(HW, 64), n_img), dim3(256), static_cast<size_t>(64) * (C + 1) * sizeof(float),
                       as_stream(stream), in, in_ld, static_cast<long long>(in_img_stride), C, HW, out);
    return check_launch("nhwc_to_nchw");
}
