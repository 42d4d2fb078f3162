// Lift-Splat frustum-to-voxel pooling for gfx950 (MI355X).
//
// Replaces the ATen call sequence of the reference's `get_geometry` / `projection_to_birds_eye_view`
// / `VoxelsSumming` (fiery/models/fiery.py:193-208, 221-273; fiery/utils/geometry.py:283-302).
//
// Design: the reference sorts 454k points per frame and prefix-sums 116 MB of
// features.  Here nothing is sorted.  A prepass turns geometry into voxel ranks and classifies every
// image *column* (fixed camera, depth, u; varying v) - for any roughly upright camera all the points of a
// column land in the same voxel.  The pooling kernel owns one (frame, channel, voxel-tile) per
// workgroup, keeps that tile of the output plane in LDS, streams the channel's feature plane exactly once
// with unit-stride wavefront loads, reduces each column in registers and retires one LDS atomic per
// column; the finished tile is written out with one coalesced pass.  HBM traffic is the algorithmic
// minimum: every feature byte once, every output byte once.
//
// This translation unit is compiled with -ffp-contract=off: the index path must reproduce ATen's CPU
// rounding (separate multiply and add, k ascending) bit for bit.
#include "common.h"
#include <fiery_gfx950.h>

#include <cmath>
#include <cstdio>
#include <type_traits>

#include <cstdlib>

#pragma clang fp contract(off)

// Row loads of the pooling kernels: plain by default.  A non-temporal hint looks right for a tensor that is read once,
// and a plain streaming sweep does gain from it - but consecutive 240-byte rows share a 64-byte sector, which a
// non-temporal load does not leave behind for the next row: +13 % HBM fetch, 229 vs 209 us for the bare read pattern
// (tools/probe/hbm_probe.hip k_read_planes, profiles/r1_s3_pool_sweeps.txt).
#ifndef POOL_EXP
#define POOL_EXP 0          // timing experiments on the compact pooling kernel (WRONG results): 1 whole units write one chunk of their plane,
#endif                      // 2 no LDS filing, 4 no many-run walk, 5 tail parts write one chunk
#ifndef FIERY_POOL_SKIP_DEAD_SLICES
#define FIERY_POOL_SKIP_DEAD_SLICES 1     // step over slices whose live word is zero (A/B switch)
#endif
#ifndef FIERY_POOL_ROW_AUX
#define FIERY_POOL_ROW_AUX 2        // cache policy bits of the row loads (buffer-load aux: 1 sc0, 2 nt, 16 sc1): non-temporal
#endif
#ifndef FIERY_POOL_NT_LOADS
#define FIERY_POOL_NT_LOADS 0
#endif

namespace fiery {
namespace {

struct GridParams {
    float ox, oy, oz;
    float rx, ry, rz;
    int nx, ny, nz;
    // Division-free forms of the quantisation, chosen per axis on the host (to_params) and bit-identical to the
    // reference's `(g - origin) / resolution` (three correctly rounded divisions per point are ~30 of the prepass's ~50
    // vector instructions per point):
    //   mode 1: the resolution is a power of two - (g - o) * (1 / r) IS the correctly rounded quotient;
    //   mode 2: a single cell along the axis (z in every shipped configuration) - the point is kept iff
    //           lo < g - o < hi, lo / hi being the largest / smallest numerators whose correctly rounded quotient is
    //           <= -1 / >= 1 (division is monotonic; found on the host by stepping through the neighbouring floats);
    //   mode 0: divide.
    int mx, my, mz;
    float ax, ay, az;      // mode 1: 1 / r        mode 2: lo
    float bx, by, bz;      //                      mode 2: hi
};

void quantise_mode(float r, int n, int* mode, float* a, float* b) {
    *mode = 0;
    *a = *b = 0.f;
    if (!(r > 0.f) || !std::isfinite(r)) return;
    int e = 0;
    if (std::frexp(r, &e) == 0.5f && std::isnormal(1.0f / r)) {
        *mode = 1;
        *a = 1.0f / r;
        return;
    }
    if (n == 1) {
        volatile float q;                                         // (volatile: the quotient is rounded to fp32 here and now)
        float lo = -r, hi = r;
        q = lo / r;
        if (q != -1.0f) return;
        for (int i = 0; i < 16; ++i) {
            const float next = std::nextafter(lo, INFINITY);
            q = next / r;
            if (!(q <= -1.0f)) break;
            lo = next;
            if (i == 15) return;
        }
        q = hi / r;
        if (q != 1.0f) return;
        for (int i = 0; i < 16; ++i) {
            const float next = std::nextafter(hi, -INFINITY);
            q = next / r;
            if (!(q >= 1.0f)) break;
            hi = next;
            if (i == 15) return;
        }
        *mode = 2;
        *a = lo;
        *b = hi;
    }
}

GridParams to_params(const fiery_bev_grid& g) {
    GridParams p{g.origin[0], g.origin[1], g.origin[2], g.resolution[0], g.resolution[1], g.resolution[2],
                 g.dim[0], g.dim[1], g.dim[2], 0, 0, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!getenv("FIERY_POOL_DIVIDE")) {                            // tests / A-B runs: the divisions everywhere
        quantise_mode(p.rx, p.nx, &p.mx, &p.ax, &p.bx);
        quantise_mode(p.ry, p.ny, &p.my, &p.ay, &p.by);
        quantise_mode(p.rz, p.nz, &p.mz, &p.az, &p.bz);
    }
    return p;
}

// ((g - origin) / resolution).long() with the reference's bounds mask (fiery.py:236-247): `.long()`
// truncates toward zero, so anything in (-1, 0) lands in cell 0 and is kept.
__device__ __forceinline__ bool quantise(float g, float origin, float res, int n, int& cell) {
    const float s = (g - origin) / res;
    // trunc(s) in [0, n)  <=>  -1 < s < n ; NaN fails both comparisons
    if (s > -1.0f && s < static_cast<float>(n)) {
        cell = static_cast<int>(s);
        return true;
    }
    cell = (s != s) ? static_cast<int>(0x80000000u)
                    : (s >= 2147483648.0f ? 0x7fffffff : (s <= -2147483648.0f ? static_cast<int>(0x80000000u)
                                                                                : static_cast<int>(s)));
    return false;
}

// The same decision and cell through the axis' division-free form (`mode`, uniform: a scalar branch); only for callers
// that do not need the cell of a point outside the grid.
__device__ __forceinline__ bool quantise_kept(float g, float origin, float res, int n, int mode, float a, float b, int& cell) {
    const float d = g - origin;
    if (mode == 2) {
        cell = 0;
        return d > a && d < b;
    }
    const float s = mode == 1 ? d * a : d / res;
    cell = static_cast<int>(s);
    return s > -1.0f && s < static_cast<float>(n);
}

__device__ __forceinline__ int voxel_rank(float gx, float gy, float gz, const GridParams& p, int* idx3) {
    int ix, iy, iz;
    if (!idx3) {
        const bool kx = quantise_kept(gx, p.ox, p.rx, p.nx, p.mx, p.ax, p.bx, ix);
        const bool ky = quantise_kept(gy, p.oy, p.ry, p.ny, p.my, p.ay, p.by, iy);
        const bool kz = quantise_kept(gz, p.oz, p.rz, p.nz, p.mz, p.az, p.bz, iz);
        return (kx && ky && kz) ? (ix * (p.ny * p.nz) + iy * p.nz + iz) : -1;
    }
    const bool kx = quantise(gx, p.ox, p.rx, p.nx, ix);
    const bool ky = quantise(gy, p.oy, p.ry, p.ny, iy);
    const bool kz = quantise(gz, p.oz, p.rz, p.nz, iz);
    idx3[0] = ix;
    idx3[1] = iy;
    idx3[2] = iz;
    return (kx && ky && kz) ? (ix * (p.ny * p.nz) + iy * p.nz + iz) : -1;
}

// ------------------------------------------------------------------------------------------------
// camera matrices: M = R . K^-1, t
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_matrix(const float* __restrict__ K, const float* __restrict__ E, float* __restrict__ out) {
    float inv[9];
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const bool canonical = K[1] == 0.f && K[3] == 0.f && K[6] == 0.f && K[7] == 0.f && K[8] == 1.f &&
                           fabsf(fx) >= fabsf(cx) && fabsf(fy) >= fabsf(cy) && fx != 0.f && fy != 0.f;
    if (canonical) {
        // What LAPACK (getrf + solve on the column-major view, no pivoting) returns for a zero-skew
        // pinhole matrix: reciprocal-scaled multiplier for the first column, a division for the second.
        const float rfx = 1.0f / fx;
        const float rfy = 1.0f / fy;
        inv[0] = rfx;  inv[1] = 0.f;  inv[2] = -(cx * rfx);
        inv[3] = 0.f;  inv[4] = rfy;  inv[5] = -(cy / fy);
        inv[6] = 0.f;  inv[7] = 0.f;  inv[8] = 1.f;
    } else {
        // general 3x3: adjugate / determinant (a few ulp from LAPACK; documented in DESIGN.md)
        const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], k = K[8];
        const float A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g;
        const float det = a * A + b * B + c * C;
        const float r = 1.0f / det;
        inv[0] = A * r;  inv[1] = -(b * k - c * h) * r;  inv[2] = (b * f - c * e) * r;
        inv[3] = B * r;  inv[4] = (a * k - c * g) * r;   inv[5] = -(a * f - c * d) * r;
        inv[6] = C * r;  inv[7] = -(a * h - b * g) * r;  inv[8] = (a * e - b * d) * r;
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            // ATen's small-matmul CPU kernel: products and sums rounded one by one, k ascending
            float acc = E[r * 4 + 0] * inv[0 * 3 + c];
            acc = acc + E[r * 4 + 1] * inv[1 * 3 + c];
            acc = acc + E[r * 4 + 2] * inv[2 * 3 + c];
            out[r * 3 + c] = acc;
        }
        out[9 + r] = E[r * 4 + 3];
    }
}


__global__ void k_camera_matrices(const float* __restrict__ intrinsics, const float* __restrict__ extrinsics,
                                  int n, float* __restrict__ cam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    camera_matrix(intrinsics + i * 9, extrinsics + i * 16, cam + i * 12);
}

// ------------------------------------------------------------------------------------------------
// camera matrices through the calibration table (fiery_amd/calibration.py): the HOST inverts (LAPACK, as the reference's
// CPU path does), keyed on the 21 words of K and [R | t] that enter the computation; the device looks the words up.
// One workgroup: a step has 54 cameras, and a launch-private miss count needs no clearing.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned calibration_hash(const unsigned* key) {
    unsigned h = 2166136261u;                                              // FNV-1a over the words, one final fold
    for (int i = 0; i < FIERY_CALIB_KEY_WORDS; ++i) h = (h ^ key[i]) * 16777619u;
    return h ^ (h >> 15);
}

__global__ __launch_bounds__(64) void k_camera_matrices_cached(const float* __restrict__ intrinsics,
                                                                const float* __restrict__ extrinsics, int n,
                                                                const unsigned* __restrict__ table, int slots,
                                                                unsigned* __restrict__ misses, int miss_capacity,
                                                                float* __restrict__ cam) {
    __shared__ int n_miss;
    if (threadIdx.x == 0) n_miss = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        unsigned key[FIERY_CALIB_KEY_WORDS];
        const unsigned* k_words = reinterpret_cast<const unsigned*>(intrinsics) + i * 9;
        const unsigned* e_words = reinterpret_cast<const unsigned*>(extrinsics) + i * 16;
        for (int w = 0; w < 9; ++w) key[w] = k_words[w];
        for (int w = 0; w < 12; ++w) key[9 + w] = e_words[w];                // the rows of [R | t]
        const unsigned h = calibration_hash(key);
        const unsigned* hit = nullptr;
        for (int probe = 0; probe < FIERY_CALIB_PROBES && !hit; ++probe) {
            const unsigned* e = table + static_cast<size_t>((h + probe) & (slots - 1)) * FIERY_CALIB_ENTRY_WORDS;
            if (e[0] == 0u) break;                                          // entries are never removed: an empty slot ends the chain
            bool same = true;
            for (int w = 0; w < FIERY_CALIB_KEY_WORDS; ++w) same = same && e[1 + w] == key[w];
            if (same) hit = e;
        }
        float* out = cam + i * 12;
        if (hit) {
            unsigned* out_words = reinterpret_cast<unsigned*>(out);
            for (int w = 0; w < 12; ++w) out_words[w] = hit[1 + FIERY_CALIB_KEY_WORDS + w];
        } else {
            camera_matrix(intrinsics + i * 9, extrinsics + i * 16, out);
            const int m = atomicAdd(&n_miss, 1);
            if (m < miss_capacity) {
                unsigned* row = misses + FIERY_CALIB_MISS_HEADER + m * FIERY_CALIB_MISS_WORDS;
                for (int w = 0; w < FIERY_CALIB_KEY_WORDS; ++w) row[w] = key[w];
                row[FIERY_CALIB_KEY_WORDS] = static_cast<unsigned>(i);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the list is rewritten by every launch; the launch number, in front of and behind the entries, tells the host
        // which launch a copy of the list belongs to and whether the copy was complete
        const unsigned launch = misses[2] + 1u;
        misses[0] = static_cast<unsigned>(min(n_miss, miss_capacity));
        misses[1] = static_cast<unsigned>(n_miss);
        misses[2] = launch;
        misses[FIERY_CALIB_MISS_HEADER + miss_capacity * FIERY_CALIB_MISS_WORDS] = launch;
    }
}

__device__ __forceinline__ void lift_point(const float* __restrict__ cam, float u, float v, float d,
                                           float& gx, float& gy, float& gz) {
    const float p0 = u * d, p1 = v * d;                      // fiery.py:202
    float a;
    a = cam[0] * p0;  a = a + cam[1] * p1;  a = a + cam[2] * d;  gx = a + cam[9];
    a = cam[3] * p0;  a = a + cam[4] * p1;  a = a + cam[5] * d;  gy = a + cam[10];
    a = cam[6] * p0;  a = a + cam[7] * p1;  a = a + cam[8] * d;  gz = a + cam[11];
}

__global__ void k_lift_geometry(const float* __restrict__ frustum, const float* __restrict__ cam,
                                int n_cam, int points_per_cam, float* __restrict__ geometry) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long total = static_cast<long long>(n_cam) * points_per_cam;
    if (i >= total) return;
    const int c = static_cast<int>(i / points_per_cam);
    const int p = static_cast<int>(i - static_cast<long long>(c) * points_per_cam);
    const float* fr = frustum + 3ll * p;
    float gx, gy, gz;
    lift_point(cam + c * 12, fr[0], fr[1], fr[2], gx, gy, gz);
    float* g = geometry + 3 * i;
    g[0] = gx;          // (plain stores: past the caches the prepass behind it took 1.1 us longer and the pooling kernel nothing less)
    g[1] = gy;
    g[2] = gz;
}

// ------------------------------------------------------------------------------------------------
// integer path
// ------------------------------------------------------------------------------------------------
__global__ void k_voxel_index(const float* __restrict__ geometry, long long n, GridParams p,
                              int* __restrict__ rank, int* __restrict__ idx) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* g = geometry + 3 * i;
    rank[i] = voxel_rank(g[0], g[1], g[2], p, idx ? idx + 3 * i : nullptr);
}

constexpr int kMaxTiles = 30;          // tile membership is a bit mask in an int
constexpr int kPackW = 10, kPackD = 10;   // list entries pack (camera, d, w) into one int

// Column descriptor, one int4 per image column (frame*camera, d, w): the column's rows as up to three runs
//   .x / .y / .z = voxel rank of rows [0, s1) / [s1, s2) / [s2, H)   (-1: those rows fall outside the grid)
//   .w = s1 | s2 << 12 | general << 24        (s1 = s2 = H: one run; s2 = H: two runs)
//   general = the column has more than three runs: its rows are then walked one by one with `rank`
// An upright camera gives single-run columns; a fraction of a degree of pitch or roll makes columns straddle
// two or three voxels (35 % / 7 % of the columns of the synthetic baseline rig).
constexpr int kSplitBits = 12;
// Compact form (`compact` != 0; grids below 65535 voxels, H < 64): 8 bytes per column,
//   .x = rankA | rankB << 16, .y = rankC | (s1 | s2 << 6 | general << 12) << 16, a rank of 0xffff = outside the grid;
// no tile masks are written (the whole-plane kernel that reads this form needs no lists).
constexpr unsigned kNoRank16 = 0xffffu;
// `occ` (optional): one byte per voxel and frame, set to 1 for every voxel some column run lands in (plain stores of
// the same value: no atomics) - the occupancy map the compact-plane kernel numbers its LDS cells from.
template <int kRows>
__global__ void k_rank_columns(const float* __restrict__ geometry, int n_fc, int D, int H, int W, GridParams p,
                               int tile_vox, int* __restrict__ rank, int4* __restrict__ coldesc,
                               int* __restrict__ colmask, int compact, unsigned char* __restrict__ occ, int n_cam,
                               long long occ_stride, unsigned* __restrict__ live, float* __restrict__ clear,
                               long long clear_floats) {
    // `clear` (optional): output planes the pooling kernel will ADD to (the units of its last, partly filled round are cut
    // into parts that meet in the output through atomics): zeroed here, a few 16-byte stores per thread, instead of by a
    // memset dispatch of its own between the two kernels (round 3: ~6 us of the op with its launch gap).
    if (clear) {
        const long long n_thr = static_cast<long long>(gridDim.x) * blockDim.x;
        const long long me = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
        if ((reinterpret_cast<uintptr_t>(clear) & 15) == 0) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long long i = me * 4; i + 3 < clear_floats; i += n_thr * 4) *reinterpret_cast<float4*>(clear + i) = z;
            if (me < (clear_floats & 3)) clear[(clear_floats & ~3ll) + me] = 0.f;
        } else {
            for (long long i = me; i < clear_floats; i += n_thr) clear[i] = 0.f;
        }
    }
    // `live` (optional, pre-zeroed): one word per (frame, camera, depth) slice, bit q set when quad q (columns 4q .. 4q+3)
    // has a point inside the grid - the compact-plane kernel does not request the rows of the others.  A workgroup's
    // columns span a few slices: their bits meet in LDS first, then one global atomic per slice and workgroup.
    __shared__ unsigned live_lds[72];                                    // 256 columns of at least 4: at most 65 slices
    const long long col_first = static_cast<long long>(blockIdx.x) * blockDim.x;
    if (live) {
        if (threadIdx.x < 72) live_lds[threadIdx.x] = 0u;
        __syncthreads();
    }
    const long long col = col_first + threadIdx.x;
    const long long n_cols = static_cast<long long>(n_fc) * D * W;
    bool any_inside = false;
    long long fd_mine = 0;
    int w_mine = 0;
    if (col < n_cols) {
    const int w = static_cast<int>(col % W);
    const long long fd = col / W;                          // (frame*camera)*D + d
    const long long base = fd * H * W + w;                 // point index of (.., h = 0, w)
    unsigned char* occ_f = occ ? occ + (fd / (static_cast<long long>(n_cam) * D)) * occ_stride : nullptr;
    fd_mine = fd;
    w_mine = w;
    int ra = -1, rb = -1, rc = -1, s1 = H, s2 = H, runs = 0, prev = 0, mask = 0;
    // kRows rows at a time: their 3 kRows coordinate loads are independent and all in flight before the run
    // bookkeeping, which is the only sequential part
    for (int h0 = 0; h0 < H; h0 += kRows) {
        float gx[kRows], gy[kRows], gz[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int h = h0 + j < H ? h0 + j : H - 1;                 // clamp: rows past the end re-read the last one
            const float* g = geometry + 3 * (base + static_cast<long long>(h) * W);
            gx[j] = g[0];
            gy[j] = g[1];
            gz[j] = g[2];
        }
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int h = h0 + j;
            if (h >= H) break;
            const int r = voxel_rank(gx[j], gy[j], gz[j], p, nullptr);
            rank[base + static_cast<long long>(h) * W] = r;
            if (h == 0 || r != prev) {
                if (runs == 0) ra = r;
                else if (runs == 1) { rb = r; s1 = h; }
                else if (runs == 2) { rc = r; s2 = h; }
                ++runs;
                if (occ_f && r >= 0) occ_f[r] = 1;
            }
            prev = r;
            if (r >= 0) {
                mask |= 1 << (r / tile_vox);
                any_inside = true;
            }
        }
    }
    const int general = (runs > 3 || H >= (1 << kSplitBits)) ? 1 : 0;
    if (compact >= 2) {
        // quad records of the compact-plane kernel (layout: see k_voxel_pool_compact); a run that does not exist is
        // stored as "none", so the reader need not look at the splits to know
        const int quad_w = W >> 2;
        const long long quad = fd * quad_w + (w >> 2);
        const int k = w & 3;
        const unsigned w16 = static_cast<unsigned>(s1) | (static_cast<unsigned>(s2) << 6) | (static_cast<unsigned>(general) << 12);
        const int vb = s1 < H ? rb : -1, vc = s2 < H ? rc : -1;
        if (compact == 2) {
            unsigned short* rec = reinterpret_cast<unsigned short*>(coldesc) + quad * 16;
            auto r16 = [](int r) { return static_cast<unsigned short>(r < 0 ? kNoRank16 : static_cast<unsigned>(r)); };
            rec[k] = static_cast<unsigned short>(w16);
            rec[4 + k] = r16(ra);
            rec[8 + k] = r16(vb);
            rec[12 + k] = r16(vc);
        } else {
            char* rec = reinterpret_cast<char*>(coldesc) + quad * 64;
            reinterpret_cast<unsigned short*>(rec)[k] = static_cast<unsigned short>(w16);
            reinterpret_cast<int*>(rec + 16)[k] = ra;
            reinterpret_cast<int*>(rec + 32)[k] = vb;
            reinterpret_cast<int*>(rec + 48)[k] = vc;
        }
    } else if (compact) {
        auto r16 = [](int r) { return r < 0 ? kNoRank16 : static_cast<unsigned>(r); };
        const unsigned w16 = static_cast<unsigned>(s1) | (static_cast<unsigned>(s2) << 6) | (static_cast<unsigned>(general) << 12);
        reinterpret_cast<int2*>(coldesc)[col] = make_int2(static_cast<int>(r16(ra) | (r16(rb) << 16)), static_cast<int>(r16(rc) | (w16 << 16)));
    } else {
        coldesc[col] = make_int4(ra, rb, rc, s1 | (s2 << kSplitBits) | (general << (2 * kSplitBits)));
        if (colmask) colmask[col] = mask | (general ? static_cast<int>(0x80000000u) : 0);      // bit 31: walk this column row by row
    }
    }   // col < n_cols
    if (live) {
        const long long fd_first = col_first / W;                     // slice of the workgroup's first column
        if (any_inside) atomicOr(&live_lds[static_cast<int>(fd_mine - fd_first)], 1u << (w_mine >> 2));
        __syncthreads();
        if (threadIdx.x < 72 && live_lds[threadIdx.x] != 0u && fd_first + threadIdx.x < static_cast<long long>(n_fc) * D)
            atomicOr(&live[fd_first + threadIdx.x], live_lds[threadIdx.x]);
    }
}

// The same prepass with FOUR lanes per column (lane 16 p + c of a wavefront takes rows [p kRpp, (p + 1) kRpp) of the
// wavefront's column c): four times the wavefronts of k_rank_columns, each with a quarter of the loads and of the
// quantisation arithmetic (three correctly rounded divisions per point), so the memory latency of one lane's rows is
// covered by other wavefronts instead of standing in a row four times (one thread per column: 2.4 wavefronts per SIMD,
// 25 us for 52 MB; this form: ~10 per SIMD).  The four lanes of a column meet through three wave exchanges: the last
// rank of the part above (is my first row a new run?), the OR of the parts' run-start bits, and the ranks at the second
// and third run start.  Everything it writes - ranks, descriptors / quad records, tile masks, occupancy bytes, live
// masks, cleared tail planes - is what k_rank_columns writes, bit for bit.  H <= 4 kRpp <= 32.
template <int kRpp>
__global__ __launch_bounds__(256) void k_rank_columns4(const float* __restrict__ geometry, int n_fc, int D, int H, int W,
                                                        GridParams p, int tile_vox, int* __restrict__ rank,
                                                        int4* __restrict__ coldesc, int* __restrict__ colmask, int compact,
                                                        unsigned char* __restrict__ occ, int n_cam, long long occ_stride,
                                                        unsigned* __restrict__ live, float* __restrict__ clear,
                                                        long long clear_floats, int rank_sparse) {
#ifndef PRE_EXP
#define PRE_EXP 0            // timing experiments (wrong results): 1 no clear, 2 no occupancy stores, 3 no record stores, 5 no geometry loads, 6 loads only
#endif
    if (clear && PRE_EXP != 1) {
        const long long n_thr = static_cast<long long>(gridDim.x) * blockDim.x;
        const long long me = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
        if ((reinterpret_cast<uintptr_t>(clear) & 15) == 0) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (long long i = me * 4; i + 3 < clear_floats; i += n_thr * 4) *reinterpret_cast<float4*>(clear + i) = z;
            if (me < (clear_floats & 3)) clear[(clear_floats & ~3ll) + me] = 0.f;
        } else {
            for (long long i = me; i < clear_floats; i += n_thr) clear[i] = 0.f;
        }
    }
    __shared__ unsigned live_lds[24];                                    // 64 columns of at least 4: at most 17 slices
    const long long col_first = static_cast<long long>(blockIdx.x) * 64;
    if (live) {
        if (threadIdx.x < 24) live_lds[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, part = lane >> 4;
    const long long col = col_first + (threadIdx.x >> 6) * 16 + (lane & 15);
    const long long n_cols = static_cast<long long>(n_fc) * D * W;
    const bool has_col = col < n_cols;
    const long long colc = has_col ? col : n_cols - 1;                  // (idle lanes shadow the last column: they take part in the exchanges)
    const int w = static_cast<int>(colc % W);
    const long long fd = colc / W;                                       // (frame*camera)*D + d
    const long long base = fd * H * W + w;                               // point index of (.., h = 0, w)
    const int h0 = part * kRpp;
    int r[kRpp];
    {
        float gx[kRpp], gy[kRpp], gz[kRpp];
#pragma unroll
        for (int j = 0; j < kRpp; ++j) {
            const int h = h0 + j < H ? h0 + j : H - 1;                   // clamp: rows past the end re-read the last one
            const float* g = geometry + 3 * (base + static_cast<long long>(h) * W);
            if (PRE_EXP == 5) {
                gx[j] = static_cast<float>(w) - 30.f + 0.01f * h;
                gy[j] = static_cast<float>(fd % 48) * 0.7f - 10.f;
                gz[j] = 0.5f;
                continue;
            }
            gx[j] = g[0];
            gy[j] = g[1];
            gz[j] = g[2];
        }
        if (PRE_EXP == 6) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < kRpp; ++j) acc += gx[j] + gy[j] + gz[j];
            if (acc == 1.2345e-30f) rank[0] = 1;
            return;
        }
#pragma unroll
        for (int j = 0; j < kRpp; ++j) {
            r[j] = voxel_rank(gx[j], gy[j], gz[j], p, nullptr);
            if (!rank_sparse && has_col && h0 + j < H) rank[base + static_cast<long long>(h0 + j) * W] = r[j];
        }
    }
    // the rank of the row above my first one: the last row of the part above (parts 1 and 3 sit 16 lanes above parts 0 and
    // 2, part 2 sits 16 lanes above part 1: lane ^ 16 and lane ^ 48)
    const int last = r[kRpp - 1];
    const int up16 = __shfl_xor(last, 16), up48 = __shfl_xor(last, 48);
    int prev = (part & 1) ? up16 : up48;
    unsigned char* occ_f = occ ? occ + (fd / (static_cast<long long>(n_cam) * D)) * occ_stride : nullptr;
    unsigned starts = 0;                                                 // bit h: row h starts a run
    int tiles = 0;
    bool any_inside = false;
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        const int h = h0 + j;
        if (h < H) {
            if (h == 0 || r[j] != prev) {
                starts |= 1u << h;
                if (PRE_EXP != 2 && occ_f && has_col && r[j] >= 0) occ_f[r[j]] = 1;
            }
            if (r[j] >= 0) {
                tiles |= 1 << (r[j] / tile_vox);
                any_inside = true;
            }
        }
        prev = r[j];
    }
    starts |= static_cast<unsigned>(__shfl_xor(static_cast<int>(starts), 16));
    starts |= static_cast<unsigned>(__shfl_xor(static_cast<int>(starts), 32));
    tiles |= __shfl_xor(tiles, 16);
    tiles |= __shfl_xor(tiles, 32);
    int inside = any_inside ? 1 : 0;
    inside |= __shfl_xor(inside, 16);
    inside |= __shfl_xor(inside, 32);
    const int runs = __popc(starts);
    const unsigned after_first = starts & (starts - 1u);                // (bit 0 is always set: row 0 starts the first run)
    const int s1 = after_first ? __ffs(static_cast<int>(after_first)) - 1 : H;
    const unsigned after_second = after_first & (after_first - 1u);
    const int s2 = after_second ? __ffs(static_cast<int>(after_second)) - 1 : H;
    // the ranks at rows 0, s1, s2: each is held by one of the four lanes, which offers rank + 1 (the others offer 0)
    int offer_a = 0, offer_b = 0, offer_c = 0;
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        const int h = h0 + j;
        if (h == 0) offer_a = r[j] + 1;
        if (h == s1) offer_b = r[j] + 1;
        if (h == s2) offer_c = r[j] + 1;
    }
    offer_a |= __shfl_xor(offer_a, 16);  offer_a |= __shfl_xor(offer_a, 32);
    offer_b |= __shfl_xor(offer_b, 16);  offer_b |= __shfl_xor(offer_b, 32);
    offer_c |= __shfl_xor(offer_c, 16);  offer_c |= __shfl_xor(offer_c, 32);
    const int ra = offer_a - 1, rb = s1 < H ? offer_b - 1 : -1, rc = s2 < H ? offer_c - 1 : -1;
    if (rank_sparse) {
        // FIERY_POOL_NO_RANKS: the ranks of the quads with a many-run column only - the compact-plane kernel walks those
        // row by row with their ranks (a quad = four neighbouring lanes of a part: lane ^ 1, lane ^ 2)
        int many = (has_col && (runs > 3 || H >= (1 << kSplitBits))) ? 1 : 0;
        many |= __shfl_xor(many, 1);
        many |= __shfl_xor(many, 2);
        if (many && has_col) {
#pragma unroll
            for (int j = 0; j < kRpp; ++j)
                if (h0 + j < H) rank[base + static_cast<long long>(h0 + j) * W] = r[j];
        }
    }
    if (PRE_EXP != 3 && has_col && part == 0) {
        const int general = (runs > 3 || H >= (1 << kSplitBits)) ? 1 : 0;
        if (compact >= 2) {
            const int quad_w = W >> 2;
            const long long quad = fd * quad_w + (w >> 2);
            const int k = w & 3;
            const unsigned w16 = static_cast<unsigned>(s1) | (static_cast<unsigned>(s2) << 6) | (static_cast<unsigned>(general) << 12);
            if (compact == 2) {
                unsigned short* rec = reinterpret_cast<unsigned short*>(coldesc) + quad * 16;
                auto r16 = [](int v) { return static_cast<unsigned short>(v < 0 ? kNoRank16 : static_cast<unsigned>(v)); };
                rec[k] = static_cast<unsigned short>(w16);
                rec[4 + k] = r16(ra);
                rec[8 + k] = r16(rb);
                rec[12 + k] = r16(rc);
            } else {
                char* rec = reinterpret_cast<char*>(coldesc) + quad * 64;
                reinterpret_cast<unsigned short*>(rec)[k] = static_cast<unsigned short>(w16);
                reinterpret_cast<int*>(rec + 16)[k] = ra;
                reinterpret_cast<int*>(rec + 32)[k] = rb;
                reinterpret_cast<int*>(rec + 48)[k] = rc;
            }
        } else if (compact) {
            auto r16 = [](int v) { return v < 0 ? kNoRank16 : static_cast<unsigned>(v); };
            const unsigned w16 = static_cast<unsigned>(s1) | (static_cast<unsigned>(s2) << 6) | (static_cast<unsigned>(general) << 12);
            // (the one-thread form stores the ranks it met second and third even when a later run replaces nothing: same here)
            reinterpret_cast<int2*>(coldesc)[col] = make_int2(static_cast<int>(r16(ra) | (r16(rb) << 16)), static_cast<int>(r16(rc) | (w16 << 16)));
        } else {
            coldesc[col] = make_int4(ra, rb, rc, s1 | (s2 << kSplitBits) | (general << (2 * kSplitBits)));
            if (colmask) colmask[col] = tiles | (general ? static_cast<int>(0x80000000u) : 0);
        }
    }
    if (live) {
        const long long fd_first = col_first / W;                        // slice of the workgroup's first column
        if (has_col && part == 0 && inside) atomicOr(&live_lds[static_cast<int>(fd - fd_first)], 1u << (w >> 2));
        __syncthreads();
        if (threadIdx.x < 24 && live_lds[threadIdx.x] != 0u && fd_first + threadIdx.x < static_cast<long long>(n_fc) * D)
            atomicOr(&live[fd_first + threadIdx.x], live_lds[threadIdx.x]);
    }
}

// The four-lane prepass as the compact-plane form of the shipped configurations meets it, with everything the general
// kernel decides at run time decided at compile time: H == 4 kRpp rows exactly, whole workgroups of columns, 32-bit offsets
// into the geometry (buffer loads: no 64-bit address arithmetic), narrow quad records, power-of-two cells along x and y and
// a single cell along z (GridParams modes 1, 1, 2).  The general kernel executes ~1,000 SCALAR instructions per wavefront
// (mode switches, row and column guards, address carries) - 38 wavefronts per CU share one scalar unit: 16 of its 20 us
// were that, with the geometry loads and every store removed it still took 17.9 us (profiles/r6_prepass_experiments.txt).
// Writes what k_rank_columns4 writes, bit for bit.
// kWide: the 64-byte quad records of grids with 65,535 voxels or more (pon_setting.yml's 400 x 200 map).
template <int kRpp, bool kWide = false>
__global__ __launch_bounds__(256) void k_rank_columns4_lean(const float* __restrict__ geometry, int geo_bytes, int D, int W, GridParams p,
                                                             int* __restrict__ rank, unsigned short* __restrict__ recs,
                                                             unsigned char* __restrict__ occ, int slices_per_frame, int occ_stride,
                                                             unsigned* __restrict__ live, float* __restrict__ clear,
                                                             long long clear_floats, int rank_sparse) {
    constexpr int H = 4 * kRpp;
    if (clear) {                                                         // (16-byte aligned: the launcher checked)
        const long long n_thr = static_cast<long long>(gridDim.x) * blockDim.x;
        const long long me = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = me * 4; i + 3 < clear_floats; i += n_thr * 4) *reinterpret_cast<float4*>(clear + i) = z;
    }
    __shared__ unsigned live_lds[24];
    const int tid = threadIdx.x;
    const int col_first = static_cast<int>(blockIdx.x) * 64;
    if (tid < 24) live_lds[tid] = 0u;
    const int lane = tid & 63, part = lane >> 4;
    const int col = col_first + (tid >> 6) * 16 + (lane & 15);
    const int w = col % W, fd = col / W;                                 // fd = (frame * camera) * D + d
    const int base = fd * (H * W) + w;                                   // point index of (.., h = 0, w)
    const int h0 = part * kRpp;
    const __amdgpu_buffer_rsrc_t geo = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(geometry), 0, geo_bytes, 0x00020000);
    const int voff = (base + h0 * W) * 12;
    const int row_bytes = W * 12;
    float gx[kRpp], gy[kRpp], gz[kRpp];
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        gx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(geo, voff, j * row_bytes, 0));
        gy[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(geo, voff + 4, j * row_bytes, 0));
        gz[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(geo, voff + 8, j * row_bytes, 0));
    }
    __syncthreads();                                                     // (live_lds zeroed; the loads are in flight)
    int r[kRpp];
    const float nxf = static_cast<float>(p.nx), nyf = static_cast<float>(p.ny);
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        const float sx = (gx[j] - p.ox) * p.ax, sy = (gy[j] - p.oy) * p.ay, dz = gz[j] - p.oz;
        const bool keep = sx > -1.0f && sx < nxf && sy > -1.0f && sy < nyf && dz > p.az && dz < p.bz;
        r[j] = keep ? static_cast<int>(sx) * p.ny + static_cast<int>(sy) : -1;
    }
    int* rank_col = rank + base + h0 * W;
    if (!rank_sparse) {
#pragma unroll
        for (int j = 0; j < kRpp; ++j) rank_col[j * W] = r[j];
    }
    const int last = r[kRpp - 1];
    const int up16 = __shfl_xor(last, 16), up48 = __shfl_xor(last, 48);
    int prev = (part & 1) ? up16 : up48;
    unsigned char* occ_f = occ + (fd / slices_per_frame) * occ_stride;
    unsigned starts = 0;                                                 // bit h: row h starts a run
    int inside = 0;
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        const bool start = (part == 0 && j == 0) || r[j] != prev;
        starts |= start ? 1u << (h0 + j) : 0u;
        if (start && r[j] >= 0) occ_f[r[j]] = 1;
        inside |= r[j] >= 0 ? 1 : 0;
        prev = r[j];
    }
    starts |= static_cast<unsigned>(__shfl_xor(static_cast<int>(starts), 16));
    starts |= static_cast<unsigned>(__shfl_xor(static_cast<int>(starts), 32));
    inside |= __shfl_xor(inside, 16);
    inside |= __shfl_xor(inside, 32);
    const int runs = __popc(starts);
    const unsigned after_first = starts & (starts - 1u);                // (bit 0 is always set: row 0 starts the first run)
    const int s1 = after_first ? __ffs(static_cast<int>(after_first)) - 1 : H;
    const unsigned after_second = after_first & (after_first - 1u);
    const int s2 = after_second ? __ffs(static_cast<int>(after_second)) - 1 : H;
    int offer_a = 0, offer_b = 0, offer_c = 0;                           // the ranks at rows 0, s1, s2 (+ 1; the other lanes offer 0)
#pragma unroll
    for (int j = 0; j < kRpp; ++j) {
        const int h = h0 + j;
        offer_a = h == 0 ? r[j] + 1 : offer_a;
        offer_b = h == s1 ? r[j] + 1 : offer_b;
        offer_c = h == s2 ? r[j] + 1 : offer_c;
    }
    offer_a |= __shfl_xor(offer_a, 16);  offer_a |= __shfl_xor(offer_a, 32);
    offer_b |= __shfl_xor(offer_b, 16);  offer_b |= __shfl_xor(offer_b, 32);
    offer_c |= __shfl_xor(offer_c, 16);  offer_c |= __shfl_xor(offer_c, 32);
    const int ra = offer_a - 1, rb = s1 < H ? offer_b - 1 : -1, rc = s2 < H ? offer_c - 1 : -1;
    const int general = runs > 3 ? 1 : 0;
    if (rank_sparse) {                                                   // (see k_rank_columns4: the many-run quads' ranks only)
        int many = general;
        many |= __shfl_xor(many, 1);
        many |= __shfl_xor(many, 2);
        if (many) {
#pragma unroll
            for (int j = 0; j < kRpp; ++j) rank_col[j * W] = r[j];
        }
    }
    if (part == 0) {
        // (one 8-byte store per lane after a 4 x 4 lane transpose of the quad's sixteen 16-bit fields measured the same as these
        // four 2-byte stores - 14.8 against 14.6 us - and was not kept)
        const int quad = fd * (W >> 2) + (w >> 2), k = w & 3;
        const unsigned short split = static_cast<unsigned short>(static_cast<unsigned>(s1) | (static_cast<unsigned>(s2) << 6) | (static_cast<unsigned>(general) << 12));
        if constexpr (kWide) {
            char* rec = reinterpret_cast<char*>(recs) + static_cast<long long>(quad) * 64;
            reinterpret_cast<unsigned short*>(rec)[k] = split;
            reinterpret_cast<int*>(rec + 16)[k] = ra;
            reinterpret_cast<int*>(rec + 32)[k] = rb;
            reinterpret_cast<int*>(rec + 48)[k] = rc;
        } else {
            unsigned short* rec = recs + quad * 16 + k;
            auto r16 = [](int v) { return static_cast<unsigned short>(v < 0 ? kNoRank16 : static_cast<unsigned>(v)); };
            rec[0] = split;
            rec[4] = r16(ra);
            rec[8] = r16(rb);
            rec[12] = r16(rc);
        }
        const int fd_first = col_first / W;                              // slice of the workgroup's first column
        if (inside) atomicOr(&live_lds[fd - fd_first], 1u << (w >> 2));
    }
    __syncthreads();
    if (tid < 24 && live_lds[tid] != 0u) atomicOr(&live[col_first / W + tid], live_lds[tid]);
}

// Ordered (ascending id) list of the work-items that touch one tile of one frame; adjacent entries are adjacent
// in memory, so the pooling kernel's wavefront loads stay unit-stride.  Items with a "general" column (more than
// three runs) go to a second list that grows down from the end of the same array; counts = {front, back}.  A work-item is a column (group = 1) or
// four adjacent columns (group = 4, W % 4 == 0): entry = camera<<20 | d<<10 | (w / group).
// grid (n_tiles, frames), 1024 threads.
__global__ __launch_bounds__(1024) void k_build_tile_lists(const int* __restrict__ colmask, int n_cam, int D, int W,
                                                           int group, int n_tiles, int* __restrict__ lists,
                                                           int* __restrict__ counts) {
    __shared__ int wave_count[16], wave_gcount[16];
    __shared__ int running, running_general;
    const int tile = blockIdx.x, f = blockIdx.y;
    const int Wg = W / group;
    const int n_items = n_cam * D * Wg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int* cm = colmask + static_cast<long long>(f) * n_items * group;
    int* out = lists + (static_cast<long long>(f) * n_tiles + tile) * n_items;
    if (threadIdx.x == 0) running = running_general = 0;
    __syncthreads();
    // four consecutive items per thread and pass: 4096 items between barriers
    constexpr int kPer = 4;
    for (int i0 = 0; i0 < n_items; i0 += 1024 * kPer) {
        const int ibase = i0 + threadIdx.x * kPer;
        bool hit[kPer], gen[kPer];
        int before = 0, wave_total = 0, gbefore = 0, gwave_total = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = ibase + k;
            hit[k] = gen[k] = false;
            if (i < n_items) {
                int mk = 0;
                for (int g = 0; g < group; ++g) mk |= cm[i * group + g];
                const bool touches = (mk >> tile) & 1;
                gen[k] = touches && mk < 0;                              // bit 31: some column needs the row walk
                hit[k] = touches && mk >= 0;
            }
            const unsigned long long lower = (1ull << lane) - 1ull;
            const unsigned long long ballot = __ballot(hit[k]);
            before += __popcll(ballot & lower);                          // hits of lower lanes come first (their items are lower)
            wave_total += __popcll(ballot);
            const unsigned long long gballot = __ballot(gen[k]);
            gbefore += __popcll(gballot & lower);
            gwave_total += __popcll(gballot);
        }
        if (lane == 0) {
            wave_count[wave] = wave_total;
            wave_gcount[wave] = gwave_total;
        }
        __syncthreads();
        int offset = running + before, goffset = running_general + gbefore;
        for (int k = 0; k < wave; ++k) {
            offset += wave_count[k];
            goffset += wave_gcount[k];
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = ibase + k;
            const int wg = i % Wg, nd = i / Wg;
            const int entry = ((nd / D) << (kPackW + kPackD)) | ((nd % D) << kPackW) | wg;
            if (hit[k]) out[offset++] = entry;
            if (gen[k]) out[n_items - 1 - goffset++] = entry;             // general items: from the back, downwards
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0, gtotal = 0;
            for (int k = 0; k < 16; ++k) {
                total += wave_count[k];
                gtotal += wave_gcount[k];
            }
            running += total;
            running_general += gtotal;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[2 * (f * n_tiles + tile)] = running;
        counts[2 * (f * n_tiles + tile) + 1] = running_general;
    }
}

// ------------------------------------------------------------------------------------------------
// pooling: one workgroup = (voxel tile, channel, frame)
// ------------------------------------------------------------------------------------------------
struct PoolStrides {
    long long f, n, d, h, w, c;
};
typedef float vf4 __attribute__((vector_size(16)));

// Accumulator cell of the LDS plane.  The default is an fp32 LDS atomic (ds_add_f32): sums of a voxel's
// columns may be added in any order, so the last bit can differ between runs.  The reproducible mode
// keeps a 64-bit fixed-point cell (2^-32 resolution, |sum| < 2^31): integer addition is associative, so
// the result is bit-identical from run to run whatever the arrival order.
template <bool kFixed> struct Cell;
template <> struct Cell<false> {
    using type = float;
    static __device__ __forceinline__ void add(float* cell, float v) { atomicAdd(cell, v); }
    static __device__ __forceinline__ float value(float c) { return c; }
};
template <> struct Cell<true> {
    using type = unsigned long long;
    static __device__ __forceinline__ void add(unsigned long long* cell, float v) {
        atomicAdd(cell, static_cast<unsigned long long>(llrintf(v * 4294967296.0f)));
    }
    static __device__ __forceinline__ float value(unsigned long long c) {
        return static_cast<float>(static_cast<double>(static_cast<long long>(c)) * (1.0 / 4294967296.0));
    }
};

// Merges consecutive contributions to the same voxel before they reach LDS: add(r, val) with r < 0 ignored.
template <bool kFixed>
struct Merge {
    int cur = -1;
    float sum = 0.f;
    __device__ __forceinline__ void add(typename Cell<kFixed>::type* plane, int v0, int v1, int r, float val) {
        if (r < v0 || r >= v1) return;                     // other tile, or outside the grid
        if (r != cur) {
            if (cur >= 0) Cell<kFixed>::add(&plane[cur - v0], sum);
            cur = r;
            sum = 0.f;
        }
        sum += val;
    }
    __device__ __forceinline__ void flush(typename Cell<kFixed>::type* plane, int v0) {
        if (cur >= 0) Cell<kFixed>::add(&plane[cur - v0], sum);
        cur = -1;
    }
};


// kVec = 4: a work-item is four adjacent columns, every row one 16-byte load; kVec = 1: one column, any strides.
// kBatch: rows of a work-item in flight at once.
// tuning aid (FIERY_POOL_PROBE): shader cycles of wave 0 of every 8th workgroup, by phase:
// {total, clear plane, load issue + wait, everything else in the item loop, merge + LDS adds, final barrier + write-out,
//  workgroups, items}
__device__ unsigned long long* g_pool_probe = nullptr;

template <int kVec, int kBatch, bool kFused, bool kFixed, bool kProbe = false, bool kPipe = false>
__global__ __launch_bounds__(kFused ? 512 : 1024) void k_voxel_pool(
    const float* __restrict__ x, PoolStrides xs,            // unfused: the lifted tensor
    const float* __restrict__ depth, const float* __restrict__ feat,   // fused: depth prob + features
    const int* __restrict__ rank, const int4* __restrict__ coldesc, const int* __restrict__ lists,
    const int* __restrict__ counts, float* __restrict__ out, int n_cam, int D, int H, int W, int C, int n_vox,
    int tile_vox, int n_tiles, int n_frames, int balanced) {
    using cell_t = typename Cell<kFixed>::type;
    HIP_DYNAMIC_SHARED(unsigned char, pool_lds)
    cell_t* plane = reinterpret_cast<cell_t*>(pool_lds);
    // unit = (tile, channel, frame).  Workgroup b is dispatched to XCD b % 8, in index order, and the length of a
    // tile's list depends on where the tile lies (the strips next to the ego vehicle hold three times the points of
    // the far ones), so the order decides the balance: the channel runs fastest - every XCD then sees the same mix of
    // tiles - and the tiles of all frames follow from the longest list to the shortest, so that the short units fill
    // the tail of the launch (longest-processing-time-first on the hardware's own greedy dispatcher).
    int tile, c, f;
    if (balanced) {
        c = blockIdx.x % C;
        const int rest = blockIdx.x / C;
        f = rest % n_frames;
        const int place = rest / n_frames;               // 0 = the frame's longest list
        int* sel = reinterpret_cast<int*>(pool_lds);
        if (static_cast<int>(threadIdx.x) < n_tiles) {
            const int* cf = counts + 2 * f * n_tiles;
            const int t = threadIdx.x;
            const int wt = cf[2 * t] + 4 * cf[2 * t + 1];      // a row-by-row item costs a few register-path items
            int before = 0;
            for (int u = 0; u < n_tiles; ++u) {
                const int wu = cf[2 * u] + 4 * cf[2 * u + 1];
                before += (wu > wt || (wu == wt && u < t)) ? 1 : 0;
            }
            if (before == place) sel[0] = t;
        }
        __syncthreads();
        tile = sel[0];
        __syncthreads();                                 // everyone has read it before the plane is cleared
    } else {
        tile = blockIdx.x % n_tiles;
        c = (blockIdx.x / n_tiles) % C;
        f = blockIdx.x / (n_tiles * C);
    }
    unsigned long long pr_t0 = 0, pr_clear = 0, pr_load = 0, pr_merge = 0, pr_items = 0;
    if constexpr (kProbe) pr_t0 = clock64();
    const int v0 = tile * tile_vox;
    const int v1 = min(v0 + tile_vox, n_vox);
    const int span = v1 - v0;
    for (int i = threadIdx.x; i < span; i += blockDim.x) plane[i] = cell_t(0);
    __syncthreads();
    if constexpr (kProbe) pr_clear = clock64() - pr_t0;

    const int Wg = W / kVec;
    const int n_items = n_cam * D * Wg;
    const int* lst = lists + (static_cast<long long>(f) * n_tiles + tile) * n_items;
    const int HW = H * W;
    const int4* cdesc = coldesc + static_cast<long long>(f) * n_items * kVec;

    // One pass over an ordered list of work-items; `dir` = +1 walks up from `first`, -1 walks down.  The items whose
    // columns all have at most three runs (the front list) take the register path; the rare ones with a column of
    // four or more runs sit in their own list (the back of the same array), so that their slow row-by-row walk is
    // executed by a few full wavefronts instead of diverging inside nearly every wavefront of the tile.
    auto run_list = [&](auto general_tag, const int* first, int dir, int count) {
        constexpr bool kGeneral = decltype(general_tag)::value;
        // Two-deep prefetch: the list entry of the item after next and the descriptors of the next item are
        // requested together with this item's rows, so no dependent (entry -> descriptor -> rows) latency is exposed.
        int e_next = 0, e_next2 = 0;
        int4 d_next[kVec];
        auto fetch_desc = [&](int e) {
            const int wg = e & ((1 << kPackW) - 1);
            const int d = (e >> kPackW) & ((1 << kPackD) - 1);
            const int cam = e >> (kPackW + kPackD);
            const int4* dp = cdesc + (static_cast<long long>(cam) * D + d) * W + wg * kVec;
#pragma unroll
            for (int k = 0; k < kVec; ++k) d_next[k] = dp[k];
        };
        const int tid = threadIdx.x, nthr = blockDim.x;
        if (tid < count) e_next = first[dir * tid];
        if (tid + nthr < count) e_next2 = first[dir * (tid + nthr)];
        if (tid < count) fetch_desc(e_next);
        for (int i = tid; i < count; i += nthr) {
            const int e = e_next;
            int4 dsc[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) dsc[k] = d_next[k];
            e_next = e_next2;
            if (i + nthr < count) fetch_desc(e_next);
            if (i + 2 * nthr < count) e_next2 = first[dir * (i + 2 * nthr)];
            const int w = (e & ((1 << kPackW) - 1)) * kVec;
            const int d = (e >> kPackW) & ((1 << kPackD) - 1);
            const int cam = e >> (kPackW + kPackD);
            // row h of this work-item lives at p + h*step (kVec floats); fused: depth row times feature row
            const float* p;
            const float* q = nullptr;
            long long step, qstep = 0;
            if (kFused) {
                p = depth + ((static_cast<long long>(f) * n_cam + cam) * D + d) * HW + w;
                step = W;
                q = feat + ((static_cast<long long>(f) * n_cam + cam) * C + c) * HW + w;
                qstep = W;
            } else {
                p = x + f * xs.f + cam * xs.n + d * xs.d + w * xs.w + c * xs.c;
                step = xs.h;
            }
            float sa[kVec], sb[kVec], sc[kVec];
            int s1[kVec], s2[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                sa[k] = sb[k] = sc[k] = 0.f;
                s1[k] = dsc[k].w & ((1 << kSplitBits) - 1);
                s2[k] = (dsc[k].w >> kSplitBits) & ((1 << kSplitBits) - 1);
            }
            // packed form of the run sums (kVec = 4, register path), two columns per register pair.  A row's membership
            // of a run is a 0/1 weight that costs one packed add with the clamp bit: sat(s1 - h) is 1 below the first
            // split and 0 from it on, sat(h + 1 - s2) the other way round for the second, and the middle run's weight is
            // what is left of 1.  A product with 0 or 1 is exact and fma(v, 1, acc) rounds like acc + v, so every run sum
            // has the bits of the plain row-by-row sum.
            constexpr bool kPacked = kVec == 4;
            constexpr int kPairs = kPacked ? 2 : 1;
            v2f first[kPairs], mid[kPairs], third[kPairs], below[kPairs], above[kPairs];
            if constexpr (kPacked) {
#pragma unroll
                for (int q = 0; q < kPairs; ++q) {
                    first[q] = mid[q] = third[q] = pk_splat(0.f);
                    below[q] = pk_make(static_cast<float>(s1[2 * q]), static_cast<float>(s1[(2 * q + 1) % kVec]));
                    above[q] = pk_make(static_cast<float>(1 - s2[2 * q]), static_cast<float>(1 - s2[(2 * q + 1) % kVec]));
                }
            }
            Merge<kFixed> merge;
            const int* rk = rank + ((static_cast<long long>(f) * n_cam + cam) * D + d) * HW + w;
            for (int h0 = 0; h0 < H; h0 += kBatch) {
                float v[kBatch][kVec];
                unsigned long long pr_a = 0;
                if constexpr (kProbe) pr_a = clock64();
                // a whole batch inside the column (the common case: the host picks a batch that divides H) needs no
                // per-row predicate and no zero fill
                auto load_rows = [&](auto full_tag) {
                    constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) {
                        const bool in = kFull || h0 + j < H;
                        if (kVec == 4) {
                            vf4 t = {0.f, 0.f, 0.f, 0.f};
                            if (in) {
                                const vf4* src = reinterpret_cast<const vf4*>(p + (h0 + j) * step);
                                // the lifted tensor is read exactly once: stream it past the caches
                                t = (kFused || !FIERY_POOL_NT_LOADS) ? *src : __builtin_nontemporal_load(src);
                            }
                            v[j][0] = t[0];  v[j][1 % kVec] = t[1];  v[j][2 % kVec] = t[2];  v[j][3 % kVec] = t[3];
                        } else {
                            v[j][0] = in ? p[(h0 + j) * step] : 0.f;
                        }
                    }
                    if (kFused) {
#pragma unroll
                        for (int j = 0; j < kBatch; ++j) {
                            const bool in = kFull || h0 + j < H;
                            if (kVec == 4) {
                                const float4 t = in ? *reinterpret_cast<const float4*>(q + (h0 + j) * qstep) : make_float4(0.f, 0.f, 0.f, 0.f);
                                v[j][0] *= t.x;  v[j][1 % kVec] *= t.y;  v[j][2 % kVec] *= t.z;  v[j][3 % kVec] *= t.w;
                            } else {
                                v[j][0] *= in ? q[(h0 + j) * qstep] : 0.f;
                            }
                        }
                    }
                };
                if (h0 + kBatch <= H) load_rows(std::true_type{});
                else load_rows(std::false_type{});
                if constexpr (kProbe) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
                    pr_load += clock64() - pr_a;
                }
                if (!kGeneral && kPacked) {
                    // 3.5 vector instructions per element (the compare / select / add form below needs 9, and its
                    // compares feed scalar ORs that stall the vector pipe); rows past the end are zeros
                    const float h0f = static_cast<float>(h0);
                    v2f b0[kPairs], a0[kPairs];
#pragma unroll
                    for (int q = 0; q < kPairs; ++q) {
                        b0[q] = below[q] - pk_splat(h0f);
                        a0[q] = above[q] + pk_splat(h0f);
                    }
#pragma unroll
                    for (int j = 0; j < kBatch; ++j)
#pragma unroll
                        for (int q = 0; q < kPairs; ++q) {
                            const v2f val = pk_make(v[j][(2 * q) % kVec], v[j][(2 * q + 1) % kVec]);
                            const v2f w_first = pk_add_sat_uniform(b0[q], pk_splat(-static_cast<float>(j)));
                            const v2f w_third = pk_add_sat_uniform(a0[q], pk_splat(static_cast<float>(j)));
                            const v2f w_mid = (pk_splat(1.f) - w_first) - w_third;
                            first[q] = pk_fma(val, w_first, first[q]);
                            mid[q] = pk_fma(val, w_mid, mid[q]);
                            third[q] = pk_fma(val, w_third, third[q]);
                        }
                } else if (!kGeneral) {
                    // each row adds to the run it belongs to (adding 0.f to the others is exact)
#pragma unroll
                    for (int j = 0; j < kBatch; ++j)
#pragma unroll
                        for (int k = 0; k < kVec; ++k) {
                            const bool first_run = h0 + j < s1[k], third_run = h0 + j >= s2[k];
                            sa[k] += first_run ? v[j][k] : 0.f;
                            sc[k] += third_run ? v[j][k] : 0.f;
                            sb[k] += (first_run || third_run) ? 0.f : v[j][k];
                        }
                } else {
                    // a column with four or more runs somewhere in this work-item: every element goes to the voxel its
                    // own rank names.  The ranks of a row are fetched like the row itself (one 16-byte load for four
                    // columns), all rows of the batch in flight together.
                    int r[kBatch][kVec];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) {
                        const bool in = h0 + j < H;
                        if (kVec == 4) {
                            const int4 t = in ? *reinterpret_cast<const int4*>(rk + (h0 + j) * W) : make_int4(-1, -1, -1, -1);
                            r[j][0] = t.x;  r[j][1 % kVec] = t.y;  r[j][2 % kVec] = t.z;  r[j][3 % kVec] = t.w;
                        } else {
                            r[j][0] = in ? rk[(h0 + j) * W] : -1;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kBatch; ++j)
#pragma unroll
                        for (int k = 0; k < kVec; ++k)
                            if (r[j][k] >= v0 && r[j][k] < v1) Cell<kFixed>::add(&plane[r[j][k] - v0], v[j][k]);
                }
            }
            if constexpr (kPacked) {
                if (!kGeneral) {
#pragma unroll
                    for (int q = 0; q < kPairs; ++q) {
                        sa[2 * q] = pk_lo(first[q]);            sa[(2 * q + 1) % kVec] = pk_hi(first[q]);
                        sb[2 * q] = pk_lo(mid[q]);              sb[(2 * q + 1) % kVec] = pk_hi(mid[q]);
                        sc[2 * q] = pk_lo(third[q]);            sc[(2 * q + 1) % kVec] = pk_hi(third[q]);
                    }
                }
            }
            unsigned long long pr_c = 0;
            if constexpr (kProbe) {
                pr_c = clock64();
                ++pr_items;
            }
            if (!kGeneral) {
#pragma unroll
                for (int k = 0; k < kVec; ++k) {
                    merge.add(plane, v0, v1, dsc[k].x, sa[k]);
                    if (s1[k] < H) merge.add(plane, v0, v1, dsc[k].y, sb[k]);
                    if (s2[k] < H) merge.add(plane, v0, v1, dsc[k].z, sc[k]);
                }
            }
            merge.flush(plane, v0);
            if constexpr (kProbe) pr_merge += clock64() - pr_c;
        }
    };
    // Software-pipelined walk of the front list (kPipe: 16-byte path, unfused, H a multiple of 2 * kBatch).  In the
    // loop above a wavefront asks for a batch of rows, waits for all of them, reduces them and only then asks for
    // more, so it has nothing in flight half of the time and the row loads see ~6 us of queueing.  Here two register
    // buffers alternate: the rows of batch k + 1 - across item boundaries, the first batch of the thread's next item -
    // are requested before batch k is reduced, so every wavefront keeps kBatch 16-byte rows per lane in flight at all
    // times.  Arithmetic and merge order are those of the plain loop: same bits.
    auto run_list_pipelined = [&](const int* first, int count) {
        static_assert(!kPipe || (kVec == 4 && !kFused), "the pipelined walk is the unfused 16-byte path");
        // Register diet (two row buffers take 56 of the 128 registers a wavefront has at four per SIMD): of the next
        // item's descriptors only the four split words are fetched ahead; the ranks of an item are requested just
        // before its last batch is reduced - ahead of the next item's rows, because loads complete in order and the
        // merge must not wait for those.
        int e_next = 0, e_next2 = 0;
        int split_next[kVec];
        auto desc_of = [&](int e) {
            const int wg = e & ((1 << kPackW) - 1);
            const int d = (e >> kPackW) & ((1 << kPackD) - 1);
            const int cam = e >> (kPackW + kPackD);
            return cdesc + (static_cast<long long>(cam) * D + d) * W + wg * kVec;
        };
        auto fetch_splits = [&](const int4* dp) {
#pragma unroll
            for (int k = 0; k < kVec; ++k) split_next[k] = reinterpret_cast<const int*>(dp + k)[3];
        };
        // A row address is (wave-uniform plane base + row * stride) + a 32-bit lane offset.  As a buffer load that is
        // descriptor + scalar offset + one vector register per item: no vector instruction is spent on addresses and
        // no 64-bit pointer is held per lane.  (The host checks that the byte offsets fit 31 bits.)
        // (sized to this (frame, channel) slab's last element: the range check covers vector + scalar offset, so a row
        // request beyond it returns zeros and touches no memory - tools/probe/buffer_oob_probe.hip)
        const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + f * xs.f + c * xs.c), 0,
            4 * static_cast<int>((n_cam - 1) * xs.n + (D - 1) * xs.d + (H - 1) * xs.h + (W - 1) * xs.w + 1), 0x00020000);
        auto rows_of = [&](int e) {
            const int w = (e & ((1 << kPackW) - 1)) * kVec;
            const int d = (e >> kPackW) & ((1 << kPackD) - 1);
            const int cam = e >> (kPackW + kPackD);
            return 4 * static_cast<int>(cam * xs.n + d * xs.d + w * xs.w);          // bytes
        };
        const int step_bytes = 4 * static_cast<int>(xs.h);
        float buf_a[kBatch][kVec], buf_b[kBatch][kVec];
        auto load_rows = [&](float (&buf)[kBatch][kVec], int p, int h0) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                // aux 2: non-temporal - the lifted tensor is read exactly once
                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rows, p, (h0 + j) * step_bytes, FIERY_POOL_NT_LOADS ? 2 : 0);
                float t[4];
                __builtin_memcpy(t, &raw, 16);
                buf[j][0] = t[0];  buf[j][1 % kVec] = t[1];  buf[j][2 % kVec] = t[2];  buf[j][3 % kVec] = t[3];
            }
        };
        const int tid = threadIdx.x, nthr = blockDim.x;
        int p_next = 0;
        const int4* dp_next = nullptr;
        if (tid < count) e_next = first[tid];
        if (tid + nthr < count) e_next2 = first[tid + nthr];
        if (tid < count) {
            dp_next = desc_of(e_next);
            fetch_splits(dp_next);
            p_next = rows_of(e_next);
            load_rows(buf_a, p_next, 0);
        }
        for (int i = tid; i < count; i += nthr) {
            const int p = p_next;
            const int4* dp = dp_next;
            constexpr int kPairs = 2;
            int s1[kVec], s2[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                s1[k] = split_next[k] & ((1 << kSplitBits) - 1);
                s2[k] = (split_next[k] >> kSplitBits) & ((1 << kSplitBits) - 1);
            }
            e_next = e_next2;
            const bool more = i + nthr < count;
            if (more) {
                dp_next = desc_of(e_next);
                fetch_splits(dp_next);
                p_next = rows_of(e_next);
            }
            if (i + 2 * nthr < count) e_next2 = first[i + 2 * nthr];
            v2f first_run[kPairs], mid_run[kPairs], third_run[kPairs], below[kPairs], above[kPairs];
#pragma unroll
            for (int q = 0; q < kPairs; ++q) {
                first_run[q] = mid_run[q] = third_run[q] = pk_splat(0.f);
                below[q] = pk_make(static_cast<float>(s1[2 * q]), static_cast<float>(s1[(2 * q + 1) % kVec]));
                above[q] = pk_make(static_cast<float>(1 - s2[2 * q]), static_cast<float>(1 - s2[(2 * q + 1) % kVec]));
            }
            auto accumulate = [&](float (&buf)[kBatch][kVec], int h0) {
                const float h0f = static_cast<float>(h0);
                v2f b0[kPairs], a0[kPairs];
#pragma unroll
                for (int q = 0; q < kPairs; ++q) {
                    b0[q] = below[q] - pk_splat(h0f);
                    a0[q] = above[q] + pk_splat(h0f);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j)
#pragma unroll
                    for (int q = 0; q < kPairs; ++q) {
                        const v2f val = pk_make(buf[j][(2 * q) % kVec], buf[j][(2 * q + 1) % kVec]);
                        const v2f w_first = pk_add_sat_uniform(b0[q], pk_splat(-static_cast<float>(j)));
                        const v2f w_third = pk_add_sat_uniform(a0[q], pk_splat(static_cast<float>(j)));
                        const v2f w_mid = (pk_splat(1.f) - w_first) - w_third;
                        first_run[q] = pk_fma(val, w_first, first_run[q]);
                        mid_run[q] = pk_fma(val, w_mid, mid_run[q]);
                        third_run[q] = pk_fma(val, w_third, third_run[q]);
                        // a row's weights are made and spent here: without the fence the scheduler computes the
                        // weights of a whole batch first and spills the row buffers to make room for them
                        if (q == kPairs - 1) __builtin_amdgcn_sched_barrier(0);
                    }
            };
            auto pin = [&]() {
#pragma unroll
                for (int q = 0; q < kPairs; ++q) pk_pin(first_run[q], mid_run[q], third_run[q]);
            };
            int rk_a[kVec], rk_b[kVec], rk_c[kVec];
            for (int h0 = 0; h0 < H; h0 += 2 * kBatch) {
                load_rows(buf_b, p, h0 + kBatch);
                accumulate(buf_a, h0);
                pin();
                if (h0 + 2 * kBatch < H) {
                    load_rows(buf_a, p, h0 + 2 * kBatch);
                } else {
#pragma unroll
                    for (int k = 0; k < kVec; ++k) {
                        const int4 t = dp[k];
                        rk_a[k] = t.x;
                        rk_b[k] = t.y;
                        rk_c[k] = t.z;
                    }
                    if (more) load_rows(buf_a, p_next, 0);
                }
                accumulate(buf_b, h0 + kBatch);
                pin();
            }
            Merge<kFixed> merge;
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                const int q = k / 2;
                const bool hi = (k & 1) != 0;
                merge.add(plane, v0, v1, rk_a[k], hi ? pk_hi(first_run[q]) : pk_lo(first_run[q]));
                if (s1[k] < H) merge.add(plane, v0, v1, rk_b[k], hi ? pk_hi(mid_run[q]) : pk_lo(mid_run[q]));
                if (s2[k] < H) merge.add(plane, v0, v1, rk_c[k], hi ? pk_hi(third_run[q]) : pk_lo(third_run[q]));
            }
            merge.flush(plane, v0);
        }
    };
    const int* cnt = counts + 2 * (f * n_tiles + tile);
    unsigned long long pr_l0 = 0, pr_l1 = 0;
    if constexpr (kProbe) pr_l0 = clock64();
    if constexpr (kPipe) run_list_pipelined(lst, cnt[0]);
    else run_list(std::false_type{}, lst, 1, cnt[0]);
    if (cnt[1] > 0) run_list(std::true_type{}, lst + n_items - 1, -1, cnt[1]);
    if constexpr (kProbe) pr_l1 = clock64();
    __syncthreads();
    float* o = out + (static_cast<long long>(f) * C + c) * n_vox + v0;
    for (int i = threadIdx.x; i < span; i += blockDim.x) o[i] = Cell<kFixed>::value(plane[i]);
    if constexpr (kProbe) {
        if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && g_pool_probe) {
            const unsigned long long t1 = clock64();
            atomicAdd(g_pool_probe + 0, t1 - pr_t0);
            atomicAdd(g_pool_probe + 1, pr_clear);
            atomicAdd(g_pool_probe + 2, pr_load);
            atomicAdd(g_pool_probe + 3, (pr_l1 - pr_l0) - pr_load - pr_merge);
            atomicAdd(g_pool_probe + 4, pr_merge);
            atomicAdd(g_pool_probe + 5, t1 - pr_l1);
            atomicAdd(g_pool_probe + 6, 1ull);
            atomicAdd(g_pool_probe + 7, pr_items);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pooling, whole-plane form: one workgroup = (channel, frame), the entire output plane in LDS
// ------------------------------------------------------------------------------------------------
// The common case (a plane of fp32 cells fits the CU's LDS: 200 x 200; 16-byte rows; unfused) needs neither tile
// lists nor tile masks: a workgroup enumerates the quads of its frame itself, reads their compact 8-byte column
// descriptors (half the bytes of the general form - descriptors are 14 % of this kernel's reads, re-fetched by every
// channel's workgroup), skips the quads that lie outside the grid without touching their rows, and parks the rare quads
// with a many-run column in an LDS queue that full wavefronts drain afterwards (when the queue is full such a quad
// is walked on the spot).  Arithmetic as in k_voxel_pool: packed run sums, neighbours merged, one LDS atomic per run.
struct FastDiv {           // n / d for n < 2^22, d < 2^18: (n * ceil(2^40 / d)) >> 40
    unsigned long long mul;
    __device__ __forceinline__ int div(int n) const { return static_cast<int>((static_cast<unsigned long long>(n) * mul) >> 40); }
};

template <int kBatch>
__global__ __launch_bounds__(1024) void k_voxel_pool_plane(const float* __restrict__ x, PoolStrides xs,
                                                           const int* __restrict__ rank, const int2* __restrict__ coldesc,
                                                           float* __restrict__ out, int n_cam, int D, int H, int W, int C,
                                                           int n_vox, int queue_cap, FastDiv div_wg, FastDiv div_d,
                                                           int tail_first, int tail_parts) {
    constexpr int kVec = 4;
    HIP_DYNAMIC_SHARED(unsigned char, plane_lds)
    float* plane = reinterpret_cast<float*>(plane_lds);
    int* queue = reinterpret_cast<int*>(plane_lds) + n_vox;          // [0] = entries pushed, then the entries
    // Units past `tail_first` are the last, partly filled round of workgroups (one workgroup per CU: 576 planes on 256
    // CUs leave 64).  Each of them is cut into `tail_parts` workgroups that take the quads block-cyclically and add
    // their partial planes to the (pre-zeroed) output, so that the whole chip works on the tail too.
    int unit = blockIdx.x, part = 0, parts = 1;
    if (unit >= tail_first) {
        const int t = unit - tail_first;
        unit = tail_first + t / tail_parts;
        part = t - (t / tail_parts) * tail_parts;
        parts = tail_parts;
    }
    const int c = unit % C;
    const int f = unit / C;
    for (int i = threadIdx.x; i < n_vox; i += blockDim.x) plane[i] = 0.f;
    if (threadIdx.x == 0) queue[0] = 0;
    __syncthreads();

    const int Wg = W / kVec;
    const int n_items = n_cam * D * Wg;
    const int HW = H * W;
    const int2* cdesc = coldesc + static_cast<long long>(f) * n_items * kVec;
    const int v0 = 0, v1 = n_vox;
    const long long step = xs.h;
    const int tid = threadIdx.x, nthr = blockDim.x;

    // a quad with a column of four or more runs: every element goes to the voxel its own rank names
    auto walk_rows = [&](int item) {
        const int nd = div_wg.div(item), wg = item - nd * Wg;
        const int cam = div_d.div(nd), d = nd - cam * D;
        const float* p = x + f * xs.f + cam * xs.n + d * xs.d + (wg * kVec) * xs.w + c * xs.c;
        const int* rk = rank + ((static_cast<long long>(f) * n_cam + cam) * D + d) * HW + wg * kVec;
        for (int h0 = 0; h0 < H; h0 += kBatch) {
            vf4 v[kBatch];
            int4 r[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const bool in = h0 + j < H;
                v[j] = vf4{0.f, 0.f, 0.f, 0.f};
                r[j] = make_int4(-1, -1, -1, -1);
                if (in) {
                    v[j] = *reinterpret_cast<const vf4*>(p + (h0 + j) * step);
                    r[j] = *reinterpret_cast<const int4*>(rk + (h0 + j) * W);
                }
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                if (r[j].x >= 0) atomicAdd(&plane[r[j].x], v[j][0]);
                if (r[j].y >= 0) atomicAdd(&plane[r[j].y], v[j][1]);
                if (r[j].z >= 0) atomicAdd(&plane[r[j].z], v[j][2]);
                if (r[j].w >= 0) atomicAdd(&plane[r[j].w], v[j][3]);
            }
        }
    };

    int4 d_next[2];                        // the four columns' descriptors of the thread's next quad
    auto fetch_desc = [&](int item) {
        const int4* dp = reinterpret_cast<const int4*>(cdesc + static_cast<long long>(item) * kVec);
        d_next[0] = dp[0];
        d_next[1] = dp[1];
    };
    const int stride = nthr * parts;
    if (tid + part * nthr < n_items) fetch_desc(tid + part * nthr);
    for (int i = tid + part * nthr; i < n_items; i += stride) {
        const int4 da = d_next[0], db = d_next[1];
        if (i + stride < n_items) fetch_desc(i + stride);
        const unsigned lo[kVec] = {static_cast<unsigned>(da.x), static_cast<unsigned>(da.z), static_cast<unsigned>(db.x),
                                   static_cast<unsigned>(db.z)};
        const unsigned hi[kVec] = {static_cast<unsigned>(da.y), static_cast<unsigned>(da.w), static_cast<unsigned>(db.y),
                                   static_cast<unsigned>(db.w)};
        int ra[kVec], rb[kVec], rc[kVec], s1[kVec], s2[kVec];
        unsigned any_general = 0;
        bool empty = true;
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const unsigned a = lo[k] & 0xffffu, b = lo[k] >> 16, cc = hi[k] & 0xffffu, w16 = hi[k] >> 16;
            ra[k] = a == kNoRank16 ? -1 : static_cast<int>(a);
            rb[k] = b == kNoRank16 ? -1 : static_cast<int>(b);
            rc[k] = cc == kNoRank16 ? -1 : static_cast<int>(cc);
            s1[k] = w16 & 63u;
            s2[k] = (w16 >> 6) & 63u;
            any_general |= w16 >> 12;
            empty = empty && a == kNoRank16 && b == kNoRank16 && cc == kNoRank16;
        }
        if (any_general) {
            const int slot = atomicAdd(&queue[0], 1);
            if (slot < queue_cap) queue[1 + slot] = i;
            else walk_rows(i);
            continue;
        }
        if (empty) continue;                                  // the whole quad lies outside the grid: its rows are not read
        const int nd = div_wg.div(i), wg = i - nd * Wg;
        const int cam = div_d.div(nd), d = nd - cam * D;
        const float* p = x + f * xs.f + cam * xs.n + d * xs.d + (wg * kVec) * xs.w + c * xs.c;
        constexpr int kPairs = 2;
        v2f first[kPairs], mid[kPairs], third[kPairs], below[kPairs], above[kPairs];
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
            first[q] = mid[q] = third[q] = pk_splat(0.f);
            below[q] = pk_make(static_cast<float>(s1[2 * q]), static_cast<float>(s1[2 * q + 1]));
            above[q] = pk_make(static_cast<float>(1 - s2[2 * q]), static_cast<float>(1 - s2[2 * q + 1]));
        }
        for (int h0 = 0; h0 < H; h0 += kBatch) {
            vf4 v[kBatch];
            auto load_rows = [&](auto full_tag) {
                constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    v[j] = vf4{0.f, 0.f, 0.f, 0.f};
                    if (kFull || h0 + j < H) {
                        const vf4* src = reinterpret_cast<const vf4*>(p + (h0 + j) * step);
                        v[j] = FIERY_POOL_NT_LOADS ? __builtin_nontemporal_load(src) : *src;   // read exactly once
                    }
                }
            };
            if (h0 + kBatch <= H) load_rows(std::true_type{});
            else load_rows(std::false_type{});
            const float h0f = static_cast<float>(h0);
            v2f b0[kPairs], a0[kPairs];
#pragma unroll
            for (int q = 0; q < kPairs; ++q) {
                b0[q] = below[q] - pk_splat(h0f);
                a0[q] = above[q] + pk_splat(h0f);
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
#pragma unroll
                for (int q = 0; q < kPairs; ++q) {
                    const v2f val = pk_make(v[j][2 * q], v[j][2 * q + 1]);
                    const v2f w_first = pk_add_sat_uniform(b0[q], pk_splat(-static_cast<float>(j)));
                    const v2f w_third = pk_add_sat_uniform(a0[q], pk_splat(static_cast<float>(j)));
                    const v2f w_mid = (pk_splat(1.f) - w_first) - w_third;
                    first[q] = pk_fma(val, w_first, first[q]);
                    mid[q] = pk_fma(val, w_mid, mid[q]);
                    third[q] = pk_fma(val, w_third, third[q]);
                }
        }
        Merge<false> merge;
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const int q = k / 2;
            const bool up = (k & 1) != 0;
            merge.add(plane, v0, v1, ra[k], up ? pk_hi(first[q]) : pk_lo(first[q]));
            if (s1[k] < H) merge.add(plane, v0, v1, rb[k], up ? pk_hi(mid[q]) : pk_lo(mid[q]));
            if (s2[k] < H) merge.add(plane, v0, v1, rc[k], up ? pk_hi(third[q]) : pk_lo(third[q]));
        }
        merge.flush(plane, v0);
    }
    __syncthreads();
    const int queued = min(queue[0], queue_cap);
    for (int qi = tid; qi < queued; qi += nthr) walk_rows(queue[1 + qi]);
    __syncthreads();
    float* o = out + (static_cast<long long>(f) * C + c) * n_vox;
    if (parts == 1) {
        for (int i = threadIdx.x; i < n_vox; i += blockDim.x) o[i] = plane[i];
    } else {
        for (int i = threadIdx.x; i < n_vox; i += blockDim.x)
            if (plane[i] != 0.f) atomicAdd(&o[i], plane[i]);
    }
}


// ------------------------------------------------------------------------------------------------
// pooling, compact-plane form: one workgroup = (channel, frame); only OCCUPIED voxels have an LDS cell
// ------------------------------------------------------------------------------------------------
// Two thirds of a frame's voxels receive no point at all (rays thin out with range: 15,037 of 40,000 voxels are hit on
// the jittered baseline rig, 12,696 of 80,000 at pon's 400 x 200), so the channel plane a workgroup accumulates needs
// 60 KB of LDS, not 160 KB: two workgroups share a CU - while one clears, drains or writes its plane out, the other
// keeps the memory pipeline busy - and grids whose dense plane does not fit a CU at all (pon) need no tiles, no lists
// and no second fetch of the rows at a tile cut.
//   * the prepass marks every voxel some column run lands in (one byte per voxel and frame, plain stores);
//   * a workgroup packs its frame's bytes into a bit map in LDS and prefix-sums the words' popcounts: the cell of
//     voxel v is  prefix[v / 32] + popcount(bits[v / 32] below bit v % 32)  - two LDS reads and a handful of integer
//     instructions per column RUN (not per point);
//   * row dealing: a wavefront takes one (camera, depth) slice at a time - H rows of W / 4 quads, contiguous in the
//     encoder's layout - and deals its rows to four 16-lane groups: lane 16 g + q reads rows g, g + 4, ... of quad q,
//     so one wavefront load covers 4 adjacent rows (960 contiguous bytes at W = 60) instead of one 240-byte row of four
//     different slices.  That makes a row's cache lines the business of ONE load instruction, and with that the
//     non-temporal hint pays (it cost 13 % extra fetch with the column-per-lane mapping, where consecutive rows of a
//     column share a line across instructions): the bare read pattern goes from 210 us (column per lane) to 197 us
//     (rows dealt) to 177 us (rows dealt, non-temporal) - profiles/r2_pool_probes.txt;
//   * rolling requests: a lane's seven rows of a slice live in seven 16-byte registers that are never idle - as soon
//     as row j of the running slice has been folded into the run sums, the same slot is asked for row j of the
//     wavefront's next slice, so a full slice per wavefront is in flight also while sums are exchanged and filed;
//   * the groups' run sums meet through two row swaps (v_permlane16_swap / v_permlane32_swap, new in gfx950:
//     vector-ALU instructions, no trip through the LDS crossbar), then group g files run g of its quad's columns;
//   * the finished plane is expanded on the way out: every voxel of the dense (X, Y) plane is stored, occupied or zero;
//   * more occupied voxels than cells (a rig unlike the one the caller sized for): the workgroup simply makes
//     several passes over its rows, one per window of cells - slower, never wrong.
// The last, partly filled round of workgroups is cut into parts that add to the pre-zeroed output, as in the
// whole-plane form.
//
// Quad records (written by the prepass for this kernel, `desc_mode` 2 / 3 of k_rank_columns): what one lane needs is one
// or two aligned loads - the four columns' split words, and the voxel of run g of each column, "no such run" already
// folded in:
//   narrow (grids below 65,535 voxels), 32 B:  u16 split[4] | u16 A[4] | u16 B[4] | u16 C[4]       (0xffff = none)
//   wide, 64 B:                                u16 split[4] | 8 B pad | i32 A[4] | i32 B[4] | i32 C[4]   (-1 = none)
//   split = s1 | s2 << 6 | many_runs << 12   (rows [0, s1) are run A, [s1, s2) run B, [s2, H) run C)
constexpr int kCompactRows = 7;            // rows per lane: the form takes H <= 4 * 7

// What a compact-form launch zeroes again, so that the next call on the workspace needs no memset (FIERY_POOL_WORKSPACE_CLEAN):
// the occupancy bytes and live masks of its frames and its counters.  Round 6: FRAME BY FRAME - the item that takes the last
// ticket of a frame class (frames f, f + 16, ...: sixteen ticket counters, so that a round's 512 tickets do not queue on one
// address) cleans those frames' 40 KB each, in the shadow of the items still running; only the last frame's is left for the end
// of the launch.  (Until then ONE workgroup - by construction the last to finish - stored the whole region, 360 KB at
// baseline.yml batch 3: 3.8 us at the tail of every launch, profiles/r6_pool_steps.txt.)
struct PoolClean {
    uint4* occ;                             // occupancy bytes of frame 0
    int occ_vec;                            // 16-byte words per frame
    unsigned* live;                         // live masks of frame 0
    int live_words;                         // words per frame
    int frames;
};
// (the tuning builds' drawn parts: one workgroup cleans everything)
template <int kThreads>
__device__ __forceinline__ void pool_clean_all(const PoolClean& cl, int* counters, int tid) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (long long i = tid; i < static_cast<long long>(cl.occ_vec) * cl.frames; i += kThreads) cl.occ[i] = z;
    for (long long i = tid; i < static_cast<long long>(cl.live_words) * cl.frames; i += kThreads) cl.live[i] = 0u;
    if (tid < 16) counters[16 + tid] = 0;
    if (tid == 16) counters[1] = 0;
    if (tid == 17) counters[0] = 0;
}
constexpr int kCompactGroupLanes = 16;     // lanes per row group: the form takes W / 4 <= 16

// kThreads: 512 (two workgroups per CU) or 1024 (one): sixteen wavefronts per CU either way; kWide: 64-byte quad records and 32-bit cell prefixes;
// kExactRows: H == 28 exactly, no row of a lane ever lies past the end of its column.
// kTuning: the instantiation with the round-4 tuning hooks (persistent grid / drawn parts / phase timeline / priorities / part
// shapes, all behind FIERY_POOL_* switches); the production instantiation has none of them - its body is round 3's plus the
// completion tickets (with the hooks compiled in, even unused, the kernel measured 6 us slower: profiles/r4_pool_decomposition.txt)
template <int kThreads, bool kWide, bool kExactRows, bool kNonTemporal, bool kTuning = false>
__global__ __launch_bounds__(kThreads, 4) void k_voxel_pool_compact(
    const float* __restrict__ x, PoolStrides xs, const int* __restrict__ rank, const void* __restrict__ quads,
    const unsigned char* __restrict__ occ, const unsigned* __restrict__ live, float* __restrict__ out,
    int* __restrict__ occupied, int n_cam, int D, int H, int W, int C, int n_vox, int n_words, int capacity, int tail_first,
    int tail_parts, int n_items, int* __restrict__ draw, long long* __restrict__ trace, int late_first, int late_prio,
    int* __restrict__ counters, PoolClean cl, int part_ranges) {
    using prefix_t = std::conditional_t<kWide, unsigned, unsigned short>;
    HIP_DYNAMIC_SHARED(unsigned char, cp_lds)
    const int n_w32 = 2 * n_words;
    unsigned* bits = reinterpret_cast<unsigned*>(cp_lds);                                    // [n_w32]
    prefix_t* prefix = reinterpret_cast<prefix_t*>(bits + n_w32);                            // [n_w32]
    // (16 bytes in front of the cells hold the item a workgroup has drawn: static LDS on top of the dynamic block would push
    // two workgroups past a CU's 160 KB)
    int* drawn = reinterpret_cast<int*>(cp_lds + ((static_cast<size_t>(n_w32) * (4 + sizeof(prefix_t)) + 15) & ~size_t(15)));
    float* plane = reinterpret_cast<float*>(drawn + 4);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // (told to the compiler in so many words: the wavefront index is the same in all lanes, so everything derived from it -
    // the slice a wavefront works on, its scalar load offsets - lives in scalar registers; left to itself the compiler
    // treats tid >> 6 as divergent and wraps every buffer load in a loop over the "different" values)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kWaves = kThreads / 64;
    // Work items: [0, tail_first) are whole (channel, frame) units, the rest are the parts of the units of the last, partly
    // filled round.  A workgroup takes items blockIdx.x, blockIdx.x + gridDim.x, ...: with a grid of one workgroup per slot
    // of the chip (the default, `persistent`) every slot does its whole units and then a part of a tail unit WITHOUT a
    // second dispatch, so the tail is cut as fine as there are slots and all of them stream to the end (576 units on 512
    // slots: one unit + one eighth each); with a grid of one workgroup per item the loop runs once.
    // `draw` (optional, zeroed by the caller): the parts of the tail units are not dealt by index but drawn from this
    // counter, so a workgroup that was slower on its whole unit (the second workgroup of a CU runs ~12 % behind the first:
    // the older wavefronts win the arbitration) takes fewer parts and all slots finish together.
    // Two workgroups share a CU and the one dispatched first runs ~12 % ahead of the other all the way (the arbiter favours
    // the older wavefronts): the later half of the first round of workgroups asks for a higher issue priority.
    if (kTuning && late_prio > 0 && static_cast<int>(blockIdx.x) >= late_first) {
        if (late_prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (late_prio == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
    int total = 0, f_have = -1, ticket = -1;
    for (int item = blockIdx.x, round = 0;; item += gridDim.x, ++round) {
    if (!kTuning && round) break;                                         // (production: one item per workgroup, no loop)
    if (kTuning && draw && item >= tail_first) {
        if (tid == 0) *drawn = tail_first + atomicAdd(draw, 1);
        __syncthreads();
        item = *drawn;
        __syncthreads();
    }
    if (item >= n_items) break;
    // tuning aid (FIERY_POOL_TRACE = address of 4 * n_items int64): the 100 MHz wall clock at the item's phases
    if (kTuning && trace && tid == 0) trace[4 * item] = wall_clock64();
    int unit = item, part = 0, parts = 1;
    if (kTuning && late_prio < 0 && unit < tail_first) unit = tail_first - 1 - unit;      // tuning (FIERY_POOL_LATE_PRIO=-1): whole units in reverse order
    if (unit >= tail_first) {
        const int t = unit - tail_first;
        unit = tail_first + t / tail_parts;
        part = t - (t / tail_parts) * tail_parts;
        parts = tail_parts;
    }
    const int c = unit % C;
    const int f = unit / C;

    const int g = lane / kCompactGroupLanes, q = lane % kCompactGroupLanes;
    const float gf = static_cast<float>(g);
    const int Wq = W >> 2;
    const bool lane_ok = q < Wq;
    const int n_slices = n_cam * D;
    const int HW = H * W;
    float* o = out + (static_cast<long long>(f) * C + c) * n_vox;
    // Row and record loads are buffer loads: descriptor (wave-uniform base of this workgroup's (frame, channel) rows / of
    // the frame's quad records) + one loop-invariant 32-bit lane offset + a scalar offset that the scalar unit advances
    // per slice and row - no 64-bit address arithmetic on the vector ALU.  A lane without work points past the
    // descriptor (reads zeros).
    constexpr int kOob = static_cast<int>(0x80000000u);
    constexpr int kRecBytes = kWide ? 64 : 32;
    // Both descriptors end with the last byte this workgroup may read (its (frame, channel) slab of rows, its frame's
    // records): the hardware's range check covers vector + scalar offset (tools/probe/buffer_oob_probe.hip), so the
    // look-ahead request past the last slice returns zeros and touches no memory.
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x + f * xs.f + c * xs.c), 0,
        4 * static_cast<int>((n_cam - 1) * xs.n + (D - 1) * xs.d + (H - 1) * xs.h + (W - 1) * xs.w + 1), 0x00020000);
    const __amdgpu_buffer_rsrc_t recs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(quads) + static_cast<long long>(f) * n_slices * Wq * kRecBytes), 0,
        n_slices * Wq * kRecBytes, 0x00020000);
    const int split_voff = lane_ok ? q * kRecBytes : kOob;
    const int rank_voff = (lane_ok && g < 3) ? q * kRecBytes + (kWide ? 16 + 16 * g : 8 + 8 * g) : kOob;
    const int row_voff = lane_ok ? 4 * static_cast<int>(g * xs.h + q * 4 * xs.w) : kOob;
    const int row4_bytes = 4 * static_cast<int>(4 * xs.h);
    auto slice_offset = [&](int s) {
        const int cam = s / D, d = s - cam * D;
        return 4 * static_cast<int>(cam * xs.n + d * xs.d);
    };
    const unsigned* live_f = live + static_cast<long long>(f) * n_slices;

    // A lane's seven rows of the running slice and the record of its four columns; both are re-requested for the
    // wavefront's next slice as soon as they have been moved aside (requests are never conditional - a slot that is
    // conditionally reloaded becomes two registers and a copy; when there is no next slice the lane offset points past
    // the descriptor, which costs no memory access).
    struct Record {
        unsigned split_lo = 0, split_hi = 0;
        int rk0 = 0, rk1 = 0, rk2 = 0, rk3 = 0;
    };
    auto request_row = [&](vf4& slot, int j, int slice_off, int lane_off) {
        const int voff = (kExactRows || 4 * j + g < H) ? lane_off : kOob;
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rows, voff, slice_off + j * row4_bytes, kNonTemporal ? FIERY_POOL_ROW_AUX : 0);
        __builtin_memcpy(&slot, &raw, 16);
    };
    // the split words of the lane's four columns and the voxels of "its" run
    auto fetch_record = [&](Record& rec, int s, bool real) {
        const int soff = real ? s * Wq * kRecBytes : 0;
        const int split_off = real ? split_voff : kOob, rank_off = real ? rank_voff : kOob;
        const auto sp = __builtin_amdgcn_raw_buffer_load_b64(recs, split_off, soff, 0);
        rec.split_lo = sp[0];
        rec.split_hi = sp[1];
        if constexpr (kWide) {
            const auto rr = __builtin_amdgcn_raw_buffer_load_b128(recs, rank_off, soff, 0);
            rec.rk0 = static_cast<int>(rr[0]);  rec.rk1 = static_cast<int>(rr[1]);
            rec.rk2 = static_cast<int>(rr[2]);  rec.rk3 = static_cast<int>(rr[3]);
        } else {
            const auto rr = __builtin_amdgcn_raw_buffer_load_b64(recs, rank_off, soff, 0);
            rec.rk0 = static_cast<int>(rr[0]);                            // two 16-bit ranks per word
            rec.rk1 = static_cast<int>(rr[1]);
        }
    };

    // The first slice's rows and record are requested BEFORE the bit map is built: they depend on nothing the set-up
    // computes, and the ~5 us of set-up (occupancy bytes -> bit map -> prefix -> cleared plane) then pass under their flight -
    // at the start of the launch every workgroup of the chip sets up at the same time and nothing streamed for that long.
    vf4 rows_in_flight[kCompactRows];
    Record record;
    const bool prefetch = !(kTuning && (draw != nullptr || part_ranges));
    bool prefetched = false;
    if (prefetch) {
        const int s0 = wave + part * kWaves;
        const bool has = s0 < n_slices;
        fetch_record(record, s0, has);
        const int off = has ? slice_offset(s0) : 0;
        const unsigned alive = has ? live_f[s0] : 0u;
#pragma unroll
        for (int j = 0; j < kCompactRows; ++j) request_row(rows_in_flight[j], j, off, ((alive >> q) & 1u) ? row_voff : kOob);
        prefetched = true;
    }

    // ---- occupancy bytes -> bit map + exclusive prefix of the words' popcounts --------------------------------
    if (f != f_have) {                                                    // (kept from the previous item of the same frame)
        f_have = f;
        const unsigned char* occ_f = occ + static_cast<long long>(f) * n_words * 64;
        const int wpt = (n_w32 + kThreads - 1) / kThreads;               // consecutive 32-voxel words per thread
        int local = 0;
        for (int k = 0; k < wpt; ++k) {
            const int w = tid * wpt + k;
            if (w >= n_w32) break;
            const uint4* src = reinterpret_cast<const uint4*>(occ_f + static_cast<long long>(w) * 32);
            const uint4 lo4 = src[0], hi4 = src[1];
            // four bytes (0 or 1) -> four bits: the products' partial terms fall on distinct bits, so nothing carries
            auto nib = [](unsigned m) { return ((m & 0x01010101u) * 0x10204080u) >> 28; };
            const unsigned b = nib(lo4.x) | nib(lo4.y) << 4 | nib(lo4.z) << 8 | nib(lo4.w) << 12 | nib(hi4.x) << 16 |
                               nib(hi4.y) << 20 | nib(hi4.z) << 24 | nib(hi4.w) << 28;
            bits[w] = b;
            local += __popc(b);
        }
        // exclusive prefix over the threads: a shuffle scan inside the wavefront, the wavefronts' totals through the LDS
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        int* wsum = reinterpret_cast<int*>(plane);                       // scratch: the plane is cleared afterwards
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = incl - local;
        total = 0;
        for (int w = 0; w < kWaves; ++w) {
            before += w < wave ? wsum[w] : 0;
            total += wsum[w];
        }
        for (int k = 0; k < wpt; ++k) {
            const int w = tid * wpt + k;
            if (w >= n_w32) break;
            prefix[w] = static_cast<prefix_t>(before);
            before += __popc(bits[w]);
        }
        __syncthreads();                                                  // scratch read, bits / prefix written
    }
    if (occupied && c == 0 && part == 0 && tid == 0) occupied[f] = total;
    if (kTuning && trace && tid == 0) trace[4 * item + 1] = wall_clock64();
    auto cell_of = [&](int r) {                                           // r: a voxel that is occupied
        return static_cast<int>(prefix[r >> 5]) + __popc(bits[r >> 5] & ((1u << (r & 31)) - 1u));
    };


    const int n_pass = total > 0 ? (total + capacity - 1) / capacity : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int lo = pass * capacity;
        const int span = min(capacity, total - lo);
        for (int i = 4 * tid; i < span; i += 4 * kThreads) *reinterpret_cast<float4*>(plane + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // (up to `capacity`, a multiple of four)
        __syncthreads();

        // one slice: `set` / `rec` hold its rows and record; they are refilled with those of slice `s_refill`
        auto process = [&](vf4 (&set)[kCompactRows], Record& rec_io, int s, int s_refill) {
            // The slice's rows have arrived: move them aside and ask for the refill slice's rows at once, all seven back
            // to back - 6,720 contiguous bytes per wavefront in one burst, before anything else is computed.  (Measured
            // alternatives, profiles/r2_pool_variants.txt: requests dealt out row by row between the arithmetic reach the
            // DRAM as seven separate visits to the same pages, microseconds apart - 434 us; two slices per wavefront in
            // flight with twelve wavefronts per CU - 341 us; this form, sixteen wavefronts with one slice ahead - 270 us.)
            vf4 cur[kCompactRows];
            const Record rec = rec_io;
            const bool refill = s_refill < n_slices;
            // A quad with a many-run column (four or more runs: a rolled camera; 1.5 % of the quads of the jittered baseline rig,
            // but one wavefront slice in ten holds one) is walked row by row - every element goes to the voxel its own rank
            // names.  The ranks are requested here, BEFORE the refill's rows, so that they come back first (loads return in
            // order), and the walk uses the rows already in `cur`.  (Until round 6 the walk came after the run sums, re-fetched
            // every row and its ranks one row at a time behind the refill - seven dependent round trips per slice: 25 us of
            // the op, found with a launch without the walk (POOL_EXP=4: 220.6 against 245.8 us).  Under FIERY_POOL_NO_RANKS the
            // prepass has written the ranks of exactly these quads.)
            const bool walk = lane_ok && ((rec.split_lo | rec.split_hi) & 0xf000f000u) != 0u;
            int4 wk[kCompactRows];
            if (walk && POOL_EXP != 4) {
                const int* rk = rank + (static_cast<long long>(f) * n_slices + s) * HW + q * 4 + g * W;
#pragma unroll
                for (int j = 0; j < kCompactRows; ++j)
                    wk[j] = (kExactRows || 4 * j + g < H) ? *reinterpret_cast<const int4*>(rk + 4 * j * W) : make_int4(-1, -1, -1, -1);
            }
            {
                const int refill_off = refill ? slice_offset(s_refill) : 0;
                // quads without a point inside the grid (a third of pon's, a twentieth of baseline's) are not fetched: the
                // prepass left one bit per quad of every slice; the word comes through the scalar cache
                const unsigned alive = refill ? live_f[s_refill] : 0u;
                const int refill_lane_off = ((alive >> q) & 1u) ? row_voff : kOob;
#pragma unroll
                for (int j = 0; j < kCompactRows; ++j) cur[j] = set[j];
#pragma unroll
                for (int j = 0; j < kCompactRows; ++j) request_row(set[j], j, refill_off, refill_lane_off);
                fetch_record(rec_io, s_refill, refill);
            }
            if (walk && POOL_EXP != 4) {
#pragma unroll
                for (int j = 0; j < kCompactRows; ++j) {
                    const int rr[4] = {wk[j].x, wk[j].y, wk[j].z, wk[j].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (rr[e] < 0) continue;
                        const int cell = cell_of(rr[e]) - lo;
                        if (static_cast<unsigned>(cell) < static_cast<unsigned>(span)) atomicAdd(&plane[cell], cur[j][e]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- what survives the row loop: two weight counters per column (packed pairs) and the cells of the runs
            // this lane files.  Row h of a column belongs to run A while s1 - h >= 1 and to run C once h + 1 - s2 >= 1:
            //   w_A = sat(below),  w_C = sat(delta - below),  below = s1 - h,  delta = s1 + 1 - s2  (a constant <= 1)
            constexpr int kPairs = 2;
            v2f below[kPairs], delta[kPairs];
            int my_cell[4];
            bool many_runs;
            {
                const unsigned w16[4] = {rec.split_lo & 0xffffu, rec.split_lo >> 16, rec.split_hi & 0xffffu, rec.split_hi >> 16};
                many_runs = lane_ok && ((w16[0] | w16[1] | w16[2] | w16[3]) >> 12) != 0;
                float s1f[4], df[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // (out-of-range lanes read zeros: s1 = s2 = 0 puts every row into run C, whose cell is "none")
                    const int s1 = static_cast<int>(w16[k] & 63u), s2 = static_cast<int>((w16[k] >> 6) & 63u);
                    s1f[k] = static_cast<float>(s1);
                    df[k] = static_cast<float>(s1 + 1 - s2);
                }
#pragma unroll
                for (int u = 0; u < kPairs; ++u) {
                    below[u] = pk_make(s1f[2 * u], s1f[2 * u + 1]) - pk_splat(gf);
                    delta[u] = pk_make(df[2 * u], df[2 * u + 1]);
                }
                int r[4];
                if constexpr (kWide) {
                    r[0] = rec.rk0;  r[1] = rec.rk1;  r[2] = rec.rk2;  r[3] = rec.rk3;
                } else {
                    const unsigned a16 = static_cast<unsigned>(rec.rk0), b16 = static_cast<unsigned>(rec.rk1);
                    const unsigned h16[4] = {a16 & 0xffffu, a16 >> 16, b16 & 0xffffu, b16 >> 16};
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = h16[k] == kNoRank16 ? -1 : static_cast<int>(h16[k]);
                }
                const bool files = lane_ok && g < 3 && !many_runs;         // (an out-of-range record reads 0 = a valid voxel)
                // looked up now, so that the LDS reads complete under the row loop
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool valid = files && r[k] >= 0;
                    const int cell = cell_of(valid ? r[k] : 0) - lo;
                    my_cell[k] = (valid && static_cast<unsigned>(cell) < static_cast<unsigned>(span)) ? cell : -1;
                }
            }
            // Six packed instructions per row and column pair: the two weights, run A's and run C's sums, the sum of
            // everything (run B is what is left of it), the counter.  No branch in here: idle lanes and many-run quads
            // (whose cells are all "none"; they were walked above) carry sums nobody files.
            v2f first[kPairs], third[kPairs], all[kPairs];
#pragma unroll
            for (int u = 0; u < kPairs; ++u) first[u] = third[u] = all[u] = pk_splat(0.f);
#pragma unroll
            for (int j = 0; j < kCompactRows; ++j) {
                const vf4 row = cur[j];
#pragma unroll
                for (int u = 0; u < kPairs; ++u) {
                    const v2f val = pk_make(row[2 * u], row[2 * u + 1]);
                    all[u] = all[u] + val;
                    const v2f w_first = pk_add_sat_uniform(below[u], pk_splat(0.f));
                    const v2f w_third = pk_sub_sat(delta[u], below[u]);
                    first[u] = pk_fma(val, w_first, first[u]);
                    third[u] = pk_fma(val, w_third, third[u]);
                    below[u] = below[u] - pk_splat(4.f);
                }
            }
            // The four row groups' sums meet as a reduce-scatter over the wavefront's four 16-lane rows, one row swap and
            // one add per step (rows_reduce_scatter3): afterwards group 0 holds the four columns' run-A totals, group 1
            // the run-B totals, group 2 the run-C totals - exactly the runs whose cells each group looked up.
            {
                float mine[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sa = (k & 1) ? pk_hi(first[k >> 1]) : pk_lo(first[k >> 1]);
                    const float sc = (k & 1) ? pk_hi(third[k >> 1]) : pk_lo(third[k >> 1]);
                    const float sb = (((k & 1) ? pk_hi(all[k >> 1]) : pk_lo(all[k >> 1])) - sa) - sc;
                    mine[k] = rows_reduce_scatter3(sa, sb, sc);
                }
                // neighbours that share a voxel are merged before they reach the LDS
                int cur = -1;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cell = my_cell[k];
                    if (cell < 0) continue;
                    if (cell != cur) {
                        if (cur >= 0 && (POOL_EXP != 2 || acc == 1.2345e-30f)) atomicAdd(&plane[cur], acc);
                        cur = cell;
                        acc = 0.f;
                    }
                    acc += mine[k];
                }
                if (cur >= 0 && (POOL_EXP != 2 || acc == 1.2345e-30f)) atomicAdd(&plane[cur], acc);
            }
        };
        // a part is a contiguous range of slices (neighbouring depths of one or two cameras: it touches a fraction of the
        // plane's cells, so its atomic write-out is short, and its rows are one stretch of memory)
        // ... when the parts are drawn (persistent workgroups); the statically dealt parts of the default launch take the
        // slices block-cyclically instead: contiguous ranges differ in how many of their quads are live (a part's row phase
        // took 29-53 us against 28-37 us, and the slowest part ends the kernel)
        // (round 6: contiguous ranges cut to EQUAL NUMBERS OF LIVE QUADS, with a third of the atomics per part, measured 12 us
        // slower still: 270.4 against 258.6 us per op, profiles/r6_pool_steps.txt)
        const bool ranges = kTuning && (draw != nullptr || part_ranges);
        const int s_lo = ranges ? static_cast<int>(static_cast<long long>(part) * n_slices / parts) : 0;
        const int s_end = ranges ? static_cast<int>(static_cast<long long>(part + 1) * n_slices / parts) : n_slices;
        const int s_first = ranges ? s_lo + wave : wave + part * kWaves, s_step = ranges ? kWaves : kWaves * parts;
        if (!prefetched) {
            const bool has = s_first < s_end;
            fetch_record(record, s_first, has);
            const int off = has ? slice_offset(s_first) : 0;
            const unsigned alive = has ? live_f[s_first] : 0u;
#pragma unroll
            for (int j = 0; j < kCompactRows; ++j) request_row(rows_in_flight[j], j, off, ((alive >> q) & 1u) ? row_voff : kOob);
        }
        prefetched = false;
        // Slices without a point inside the grid (a camera that looks away from it: a third of pon_setting.yml's slices, a few of
        // baseline's) are stepped over - their live word is zero (a wave-uniform scalar load); only a unit's first slice is taken
        // as it comes (it was requested before this loop).
        auto next_live = [&](int s) {
#if FIERY_POOL_SKIP_DEAD_SLICES
            while (s < s_end && live_f[s] == 0u) s += s_step;
#endif
            return s < s_end ? s : n_slices;
        };
        for (int s = s_first; s < s_end;) {
            const int s_next = next_live(s + s_step);
            process(rows_in_flight, record, s, s_next);
            s = s_next;
        }
        __syncthreads();
        if (kTuning && trace && tid == 0) trace[4 * item + 2] = wall_clock64();
        // this item has read the last thing it needs from the region the call clears (its live masks): it takes its ticket now,
        // so that the atomic's round trip passes under the write-out instead of holding the slot afterwards
        // (not when parts are drawn: the draw counter lives in the same region and is read until the last workgroup leaves)
        // (sixteen counters, dealt by frame: the workgroups of a round finish within microseconds of each other and an atomic on
        // ONE address is served every ~0.1 us - 512 of them held every slot for its share of 50 us)
        if (counters && !(kTuning && draw) && pass == n_pass - 1 && tid == 0) ticket = atomicAdd(counters + 16 + (f & 15), 1);
        // ---- expand the window of cells into the dense plane --------------------------------------------------
        if (parts == 1 && n_pass == 1 && (n_vox & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            // one pass, whole units: every voxel gets its cell's sum or a zero - four voxels (one nibble of a bit word) per
            // thread and 16-byte store; their cells are consecutive
            for (int v0 = 4 * tid; v0 < (POOL_EXP == 1 ? min(n_vox, 4 * kThreads) : n_vox); v0 += 4 * kThreads) {
                const unsigned wbits = bits[v0 >> 5];
                const unsigned nib = (wbits >> (v0 & 31)) & 15u;
                int cell = static_cast<int>(prefix[v0 >> 5]) + __popc(wbits & ((1u << (v0 & 31)) - 1u));
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nib & 1u) v.x = plane[cell++];
                if (nib & 2u) v.y = plane[cell++];
                if (nib & 4u) v.z = plane[cell++];
                if (nib & 8u) v.w = plane[cell];
                *reinterpret_cast<float4*>(o + v0) = v;            // (plain: non-temporal stores measured 1.7 us slower per op, round 6)
            }
        } else
        for (int v0 = tid; v0 < (POOL_EXP == 5 ? min(n_vox, kThreads) : n_vox); v0 += kThreads) {
            const unsigned wbits = bits[v0 >> 5];
            const bool hit = (wbits >> (v0 & 31)) & 1u;
            const int cell = static_cast<int>(prefix[v0 >> 5]) + __popc(wbits & ((1u << (v0 & 31)) - 1u)) - lo;
            const bool mine = hit && static_cast<unsigned>(cell) < static_cast<unsigned>(span);
            if (parts == 1) {
                if (mine) o[v0] = plane[cell];
                else if (!hit && pass == 0) o[v0] = 0.f;
            } else if (mine) {
                const float val = plane[cell];
                if (val != 0.f) atomicAdd(&o[v0], val);
            }
        }
        __syncthreads();                                                  // the plane is cleared again by the next pass
    }
    if (kTuning && trace && tid == 0) trace[4 * item + 3] = wall_clock64();
    // The item that takes the last ticket of its frame class leaves those frames' occupancy bytes and live masks as the launch
    // found them - all zero - and reports to the top counter; the class that completes that one resets it: the next call on this
    // workspace can skip its memset dispatch (FIERY_POOL_WORKSPACE_CLEAN).  Every other item of the frames had read what it
    // needed from them before it took its ticket.
    if (counters && !(kTuning && draw)) {
        const int cls = f & 15;
        if (tid == 0) {
            int expect = 0;                                               // items of the frames cls, cls + 16, ...
            for (int ff = cls; ff < cl.frames; ff += 16) {
                const int whole = min(max(tail_first - ff * C, 0), C);   // units of the frame that are not tail units
                expect += whole + (C - whole) * tail_parts;
            }
            *drawn = ticket == expect - 1;
        }
        __syncthreads();
        if (*drawn) {
            const uint4 z = {0u, 0u, 0u, 0u};
            for (int ff = cls; ff < cl.frames; ff += 16) {
                uint4* o4 = cl.occ + static_cast<long long>(ff) * cl.occ_vec;
                for (int i = tid; i < cl.occ_vec; i += kThreads) o4[i] = z;
                unsigned* lv = cl.live + static_cast<long long>(ff) * cl.live_words;
                for (int i = tid; i < cl.live_words; i += kThreads) lv[i] = 0u;
            }
            if (tid == 0) {
                counters[16 + cls] = 0;
                if (atomicAdd(counters + 1, 1) == min(cl.frames, 16) - 1) {
                    counters[1] = 0;
                    counters[0] = 0;
                }
            }
        }
        __syncthreads();                                                  // (`drawn` is written again by the next item)
    }
    }   // items
    if (kTuning && counters && draw) {                                    // drawn parts: one ticket per workgroup, after its last draw
        __syncthreads();
        if (tid == 0) *drawn = atomicAdd(counters + 1, 1);
        __syncthreads();
        if (*drawn == static_cast<int>(gridDim.x) - 1) pool_clean_all<kThreads>(cl, counters, tid);
    }
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_camera_matrices(const float* intrinsics, const float* extrinsics, int n_cameras,
                                     float* cam, fiery_stream_t stream) {
    FIERY_REQUIRE(intrinsics && extrinsics && cam && n_cameras > 0, "camera_matrices: null pointer or n <= 0");
    hipLaunchKernelGGL(k_camera_matrices, dim3(ceil_div(n_cameras, 64)), dim3(64), 0, as_stream(stream),
                       intrinsics, extrinsics, n_cameras, cam);
    return check_launch("camera_matrices");
}

extern "C" int fiery_camera_matrices_cached(const float* intrinsics, const float* extrinsics, int n_cameras,
                                            const uint32_t* table, int table_slots, uint32_t* misses, int miss_capacity,
                                            float* cam, fiery_stream_t stream) {
    FIERY_REQUIRE(intrinsics && extrinsics && cam && table && misses && n_cameras > 0,
                  "camera_matrices_cached: null pointer or n <= 0");
    FIERY_REQUIRE(table_slots >= 64 && (table_slots & (table_slots - 1)) == 0 && miss_capacity > 0,
                  "camera_matrices_cached: the table needs a power-of-two number of slots >= 64 and a miss list");
    hipLaunchKernelGGL(k_camera_matrices_cached, dim3(1), dim3(64), 0, as_stream(stream), intrinsics, extrinsics, n_cameras,
                       table, table_slots, misses, miss_capacity, cam);
    return check_launch("camera_matrices_cached");
}

extern "C" int fiery_lift_geometry(const float* frustum, const float* cam, int n_cameras, int D, int H, int W,
                                   float* geometry, fiery_stream_t stream) {
    FIERY_REQUIRE(frustum && cam && geometry, "lift_geometry: null pointer");
    FIERY_REQUIRE(n_cameras > 0 && D > 0 && H > 0 && W > 0, "lift_geometry: bad shape");
    const int pts = D * H * W;
    const long long total = static_cast<long long>(n_cameras) * pts;
    hipLaunchKernelGGL(k_lift_geometry, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), frustum, cam,
                       n_cameras, pts, geometry);
    return check_launch("lift_geometry");
}

extern "C" int fiery_voxel_index(const float* geometry, int64_t n_points, const fiery_bev_grid* grid, int32_t* rank,
                                 int32_t* idx, fiery_stream_t stream) {
    FIERY_REQUIRE(geometry && grid && rank, "voxel_index: null pointer");
    FIERY_REQUIRE(n_points >= 0, "voxel_index: negative size");
    if (n_points == 0) return FIERY_OK;
    hipLaunchKernelGGL(k_voxel_index, dim3(ceil_div(n_points, 256)), dim3(256), 0, as_stream(stream), geometry,
                       static_cast<long long>(n_points), to_params(*grid), rank, idx);
    return check_launch("voxel_index");
}

namespace {

struct PoolPlan {
    int n_vox, tile, n_tiles, n_words;
    size_t lds, off_coldesc, off_colmask, off_counts, off_lists, off_occ, off_live, off_occupied, total;
};

// LDS tile of the output plane (see the default below); the tile grows when the grid would otherwise need more
// than kMaxTiles tiles.
int plan_pool(int frames, int n_cam, int D, int H, int W, long long n_vox_ll, int requested, bool fixed, bool fused,
              PoolPlan* pl) {
    FIERY_REQUIRE(n_vox_ll > 0 && n_vox_ll < (1ll << 30), "voxel_pool: bad grid size");
    FIERY_REQUIRE(W < (1 << kPackW) && D < (1 << kPackD) && n_cam < (1 << (31 - kPackW - kPackD)),
                  "voxel_pool: feature map too large for the column encoding (W, D < 1024, cameras < 2048)");
    const int cell_bytes = fixed ? 8 : 4;
    const int cap = (fused ? 81920 : 160000) / cell_bytes;     // the fused form runs 512-thread workgroups at most
    pl->n_vox = static_cast<int>(n_vox_ll);
    // default: the whole plane when it fits the CU's LDS (200 x 200 fp32 cells = 160,000 B, one 1024-thread workgroup
    // per CU) - no row of a camera is then cut between two workgroups, which costs 12 % of the HBM reads with two
    // tiles (profiles/r1_s3_pool_sweeps.txt); otherwise 80 KiB tiles, two 512-thread workgroups per CU
    int tile = requested > 0 ? requested : (static_cast<long long>(n_vox_ll) * cell_bytes <= cap * cell_bytes && !fused ? static_cast<int>(n_vox_ll) : 81920 / cell_bytes);
    if (tile > cap) tile = cap;
    if (tile > pl->n_vox) tile = pl->n_vox;
    if (ceil_div(pl->n_vox, tile) > kMaxTiles) tile = ceil_div(pl->n_vox, kMaxTiles);
    FIERY_REQUIRE(tile <= cap, "voxel_pool: a %d-voxel grid needs more than %d LDS tiles", pl->n_vox, kMaxTiles);
    pl->tile = tile;
    pl->n_tiles = ceil_div(pl->n_vox, tile);
    pl->lds = static_cast<size_t>(tile) * cell_bytes;
    const size_t points = static_cast<size_t>(frames) * n_cam * D * H * W;
    const size_t cols = static_cast<size_t>(frames) * n_cam * D * W;
    auto align = [](size_t v) { return (v + 255) / 256 * 256; };
    pl->off_coldesc = align(points * 4);
    pl->off_colmask = align(pl->off_coldesc + cols * sizeof(int4));
    pl->off_counts = align(pl->off_colmask + cols * 4);
    pl->off_lists = align(pl->off_counts + static_cast<size_t>(frames) * pl->n_tiles * 2 * 4);
    // compact-plane form: one occupancy byte per voxel and frame (rounded up to whole 64-voxel words), and the
    // number of occupied voxels of every frame (int32), which the kernel leaves for callers that size its LDS plane
    pl->n_words = ceil_div(pl->n_vox, 64);
    pl->off_occ = align(pl->off_lists + cols * pl->n_tiles * sizeof(int));
    pl->off_live = align(pl->off_occ + static_cast<size_t>(frames) * pl->n_words * 64);     // (cleared together with the bytes)
    // (+ 256 B: the completion tickets / the counter the tuning builds draw the parts of the tail units from; cleared with the masks)
    pl->off_occupied = align(pl->off_live + static_cast<size_t>(frames) * n_cam * D * 4 + 256);
    pl->total = align(pl->off_occupied + static_cast<size_t>(frames) * 4);
    return FIERY_OK;
}

}  // namespace

extern "C" size_t fiery_voxel_pool_workspace_bytes(int frames, int n_cameras, int D, int H, int W, int n_voxels,
                                                   int tile_voxels, uint32_t flags) {
    PoolPlan pl;
    if (frames <= 0 || n_cameras <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    // sized for the fused form too (its tile cap is the smaller one, so it never needs fewer list slots)
    if (plan_pool(frames, n_cameras, D, H, W, n_voxels, tile_voxels, (flags & FIERY_POOL_DETERMINISTIC) != 0, true, &pl)) return 0;
    return pl.total;
}

extern "C" size_t fiery_voxel_pool_occupied_offset(int frames, int n_cameras, int D, int H, int W, int n_voxels,
                                                   int tile_voxels, uint32_t flags) {
    PoolPlan pl;
    if (frames <= 0 || n_cameras <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (plan_pool(frames, n_cameras, D, H, W, n_voxels, tile_voxels, (flags & FIERY_POOL_DETERMINISTIC) != 0, true, &pl)) return 0;
    return pl.off_occupied;
}

namespace {

int pool_common(bool fused, const float* x, const int64_t* xs, const float* depth, const float* feat,
                const float* geometry, int frames, int n_cam, int D, int H, int W, int C,
                const fiery_bev_grid* grid, float* out, void* workspace, size_t ws_bytes, int tile_voxels,
                uint32_t flags, fiery_stream_t stream) {
    FIERY_REQUIRE(geometry && grid && out && workspace, "voxel_pool: null pointer");
    FIERY_REQUIRE(frames > 0 && n_cam > 0 && D > 0 && H > 0 && W > 0 && C > 0, "voxel_pool: bad shape");
    FIERY_REQUIRE(grid->dim[2] == 1,
                  "voxel_pool: bev_dimension[2] = %d; the reference only supports one z cell "
                  "(fiery/models/fiery.py:268-271)", grid->dim[2]);
    FIERY_REQUIRE(grid->dim[0] > 0 && grid->dim[1] > 0, "voxel_pool: empty grid");
    FIERY_REQUIRE((flags & ~(FIERY_POOL_DETERMINISTIC | FIERY_POOL_WORKSPACE_CLEAN | FIERY_POOL_NO_RANKS)) == 0, "voxel_pool: unknown flags 0x%x", flags);
    const bool fixed = (flags & FIERY_POOL_DETERMINISTIC) != 0;
    PoolPlan pl;
    int rc = plan_pool(frames, n_cam, D, H, W, static_cast<long long>(grid->dim[0]) * grid->dim[1], tile_voxels, fixed, fused, &pl);
    if (rc) return rc;
    if (ws_bytes < pl.total) return fail(FIERY_ENOMEM, "voxel_pool: workspace %zu B < required %zu B", ws_bytes, pl.total);
    if (fused) {
        FIERY_REQUIRE(depth && feat, "lift_splat: null pointer");
    } else {
        FIERY_REQUIRE(x && xs, "voxel_pool: null pointer");
    }
    char* ws = static_cast<char*>(workspace);
    int* rank = reinterpret_cast<int*>(ws);
    int4* coldesc = reinterpret_cast<int4*>(ws + pl.off_coldesc);
    int* colmask = reinterpret_cast<int*>(ws + pl.off_colmask);
    int* counts = reinterpret_cast<int*>(ws + pl.off_counts);
    int* lists = reinterpret_cast<int*>(ws + pl.off_lists);
    const long long n_cols_all = static_cast<long long>(frames) * n_cam * D * W;
    hipStream_t s = as_stream(stream);
    // workgroup size follows the tile: 256 threads per 40 KiB of LDS keeps 16 wavefronts per CU whatever the tile
    const int threads = pl.lds <= 40960 ? 256 : (pl.lds <= 81920 ? 512 : 1024);
    if (getenv("FIERY_POOL_VERBOSE"))
        fprintf(stderr, "voxel_pool plan: frames=%d C=%d tile=%d x%d, %d threads, %zu B LDS\n", frames, C, pl.tile, pl.n_tiles,
                threads, pl.lds);
    PoolStrides st{0, 0, 0, 0, 0, 0};
    if (!fused) st = PoolStrides{xs[0], xs[1], xs[2], xs[3], xs[4], xs[5]};
    // 16-byte path: rows of four columns must be contiguous and 16-byte aligned
    bool quads = (W % 4 == 0);
    if (fused) {
        quads = quads && aligned16(depth) && aligned16(feat);
    } else {
        quads = quads && aligned16(x) && st.w == 1 && st.f % 4 == 0 && st.n % 4 == 0 && st.d % 4 == 0 && st.h % 4 == 0 &&
                st.c % 4 == 0;
    }
    if (const char* forced = getenv("FIERY_POOL_VEC")) quads = quads && atoi(forced) == 4;      // tuning / tests
    // the whole-plane kernel (compact descriptors, no lists): fp32 cells, unfused, one tile, a grid its 16-bit ranks reach
    const long long n_quads = static_cast<long long>(n_cam) * D * (W / 4);
    bool plane_form = quads && !fused && !fixed && pl.n_tiles == 1 && pl.n_vox < 65535 && H < 64 && n_quads < (1ll << 22) &&
                      W / 4 < (1 << 18) && D < (1 << 18) && static_cast<long long>(C) * frames < (1ll << 31) &&
                      !getenv("FIERY_POOL_PROBE");
    if (const char* forced = getenv("FIERY_POOL_PLANE")) plane_form = plane_form && atoi(forced) != 0;   // tuning / A-B runs
    // The compact-plane kernel (LDS cells for occupied voxels only; see k_voxel_pool_compact) - the default for the
    // unfused 16-byte path whenever a slice fits its row dealing: any grid whose occupancy bit map plus a useful number
    // of cells fit a CU's LDS, dense plane or not.  `tile_voxels` > 0 asks for that many cells (a caller that has read
    // the occupied counts of an earlier call from the workspace sizes the plane to its rig: fewer cells = more
    // workgroups per CU); 0 = as many as let two workgroups share a CU.
    int cp_cells = 0, cp_threads = 0;
    size_t cp_lds = 0;
    bool compact_form = quads && !fused && !fixed && H <= 4 * kCompactRows && W / 4 <= kCompactGroupLanes &&
                        static_cast<long long>(C) * frames < (1ll << 30) && pl.n_vox < (1 << 24) && !getenv("FIERY_POOL_PROBE");
    if (compact_form) {
        // its buffer loads address a (frame, channel) set of rows with 31-bit byte offsets
        const long long reach = 4 * ((n_cam - 1) * st.n + (D - 1) * st.d + (H - 1) * st.h + (W - 1) * st.w + 4);
        compact_form = st.n >= 0 && st.d >= 0 && st.h >= 0 && reach < (1ll << 31) &&
                       static_cast<long long>(n_cam) * D * W * 16 < (1ll << 31);
    }
    if (const char* forced = getenv("FIERY_POOL_COMPACT")) compact_form = compact_form && atoi(forced) != 0;   // tuning / A-B runs
    if (compact_form) {
        // LDS: bit map (4 B per 32 voxels) + cell prefixes (2 B, or 4 B for grids of 65,535 voxels or more), then the cells
        const long long fixed_bytes = ((static_cast<long long>(pl.n_words) * 2 * (4 + (pl.n_vox >= 65535 ? 4 : 2))) + 15) / 16 * 16 + 16;
        const long long cells_max = (163840 - fixed_bytes) / 4;
        const long long cells_two = (81920 - fixed_bytes) / 4;
        long long cells = tile_voxels > 0 ? tile_voxels : cells_two;
        if (const char* forced = getenv("FIERY_POOL_CELLS")) cells = atoll(forced);              // tuning / tests
        if (cells > pl.n_vox) cells = pl.n_vox;
        if (cells < 576) cells = 576;                                    // the prefix scan borrows the plane as scratch
        if (cells > cells_max) cells = cells_max;
        if (cells_max < 1088) compact_form = false;                      // the bit map alone fills the LDS: tiled kernel
        cp_cells = static_cast<int>(cells);
        cp_lds = static_cast<size_t>(fixed_bytes + cells * 4);
        cp_threads = 2 * cp_lds <= 163840 ? 512 : 1024;                  // two workgroups per CU, or one: 16 wavefronts either way
    }
    if (compact_form) plane_form = false;
    bool four_lanes = H <= 32;                // four lanes per column in the prepass (see k_rank_columns4)
    if (const char* forced = getenv("FIERY_POOL_PREPASS_LANES")) four_lanes = four_lanes && atoi(forced) == 4;   // tuning / A-B runs
    const bool no_ranks = compact_form && four_lanes && (flags & FIERY_POOL_NO_RANKS) != 0;
    const bool compact_desc = plane_form;
    const bool wide_records = pl.n_vox >= 65535;
    const int desc_mode = compact_form ? (wide_records ? 3 : 2) : (plane_form ? 1 : 0);
    const long long occ_stride = static_cast<long long>(pl.n_words) * 64;
    unsigned char* occ = reinterpret_cast<unsigned char*>(ws + pl.off_occ);
    int* occupied = reinterpret_cast<int*>(ws + pl.off_occupied);
    unsigned* live = reinterpret_cast<unsigned*>(ws + pl.off_live);
    int* counter_block = reinterpret_cast<int*>(ws + pl.off_occupied - 256);    // 64 ints: [0] parts drawn (tuning), [1] top counter, [16, 32) completion tickets
    // occupancy bytes + live masks + counters: cleared here unless the caller vouches that the region is as the library
    // left it (the compact form's last workgroup re-zeroes it) or as a zero-filled allocation
    if (compact_form && !(flags & FIERY_POOL_WORKSPACE_CLEAN) && hipMemsetAsync(occ, 0, pl.off_occupied - pl.off_occ, s) != hipSuccess)
        return fail(FIERY_ELAUNCH, "voxel_pool: cannot clear the occupancy map");
    // compact form: the units of the last, partly filled round are cut into parts (below); the planes they add to are zeroed
    // by the prepass
    int cp_tail = 0, cp_parts = 1, cp_slots = 0;
    // FIERY_POOL_PERSISTENT=1: one workgroup per slot of the chip, each taking its items in turn (see the kernel)
    // (measured, profiles/r4_pool_decomposition.txt: the per-item grid with the tail units cut in four is as fast or faster
    // than persistent workgroups with drawn parts - a part's fixed costs outweigh the better balance - so this is opt-in)
    bool persistent = false;
    if (const char* forced = getenv("FIERY_POOL_PERSISTENT")) persistent = atoi(forced) != 0;   // tuning / A-B runs
    if (compact_form) {
        const int n_units = C * frames;
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
        const int slots = n_cu * (cp_threads == 512 ? 2 : 1);
        cp_slots = slots;
        cp_tail = n_units % slots;
        if (cp_tail > 0) cp_parts = slots / cp_tail;
        if (n_units < slots) cp_parts = 1;                               // a single, partly filled round: nothing to balance
        if (const char* forced = getenv("FIERY_POOL_TAIL_PARTS")) {                           // tuning / tests
            cp_parts = atoi(forced);
            if (cp_parts > 1 && cp_tail == 0) cp_tail = n_units < 3 ? n_units : 3;
        }
        // (a part pays the bit-map set-up again: with a dispatch per part 4 measured best; a persistent workgroup goes on
        // to its part straight from its unit, and there as many parts as fill every slot are best)
        if (cp_parts > 4 && !persistent && !getenv("FIERY_POOL_TAIL_PARTS")) cp_parts = 4;
        // persistent workgroups draw the parts from a counter: finer parts (about three slices per wavefront) even out the
        // workgroups' different speeds
        if (persistent && cp_tail > 0 && cp_parts > 1 && !getenv("FIERY_POOL_TAIL_PARTS")) {
            const int per_part = 3 * (cp_threads / 64);
            cp_parts = (n_cam * D + per_part - 1) / per_part;
            if (cp_parts < 2) cp_parts = 2;
        }
        if (cp_parts > 16) cp_parts = 16;
        if (cp_parts < 2) {
            cp_parts = 1;
            cp_tail = 0;
        }
    }
    float* clear_ptr = cp_tail > 0 ? out + static_cast<long long>(C * frames - cp_tail) * pl.n_vox : nullptr;
    const long long clear_floats = static_cast<long long>(cp_tail) * pl.n_vox;
    {
        int rows = 8;                                                    // rows of a column in flight in the prepass
        if (const char* forced = getenv("FIERY_POOL_PREPASS_ROWS")) rows = atoi(forced);         // tuning
        unsigned char* occ_arg = compact_form ? occ : nullptr;
        int* mask_arg = (plane_form || compact_form) ? nullptr : colmask;
        const dim3 pgrid(ceil_div(n_cols_all, 256));
        // four lanes per column (see k_rank_columns4) whenever a column's rows fit four parts of 7 or 8 (its run-start bits
        // are one 32-bit word)
        const bool four = four_lanes;
        const dim3 pgrid4(ceil_div(n_cols_all, 64));
        const GridParams gp = to_params(*grid);
        // the lean form of the four-lane prepass (see the kernel): the compact form's narrow records, H = 28 exactly, whole
        // workgroups of columns, a geometry tensor of less than 2 GiB, division-free quantisation along all three axes
        const long long geo_bytes = static_cast<long long>(frames) * n_cam * D * H * W * 12;
        bool lean = four && (desc_mode == 2 || desc_mode == 3) && H == 28 && n_cols_all % 64 == 0 && geo_bytes < (1ll << 31) && gp.mx == 1 && gp.my == 1 &&
                    gp.mz == 2 && gp.nz == 1 && occ_stride * frames < (1ll << 31) && (!clear_ptr || aligned16(clear_ptr)) && clear_floats % 4 == 0 && W % 4 == 0 && W >= 4;
        if (const char* forced = getenv("FIERY_POOL_PREPASS_LEAN")) lean = lean && atoi(forced) != 0;           // tuning / A-B runs
        if (lean && desc_mode == 3)
            hipLaunchKernelGGL((k_rank_columns4_lean<7, true>), pgrid4, dim3(256), 0, s, geometry, static_cast<int>(geo_bytes), D, W, gp, rank,
                               reinterpret_cast<unsigned short*>(coldesc), occ, n_cam * D, static_cast<int>(occ_stride), live, clear_ptr,
                               clear_floats, no_ranks ? 1 : 0);
        else if (lean)
            hipLaunchKernelGGL((k_rank_columns4_lean<7>), pgrid4, dim3(256), 0, s, geometry, static_cast<int>(geo_bytes), D, W, gp, rank,
                               reinterpret_cast<unsigned short*>(coldesc), occ, n_cam * D, static_cast<int>(occ_stride), live, clear_ptr,
                               clear_floats, no_ranks ? 1 : 0);
        else if (four && H <= 28)
            hipLaunchKernelGGL((k_rank_columns4<7>), pgrid4, dim3(256), 0, s, geometry, frames * n_cam, D, H, W, to_params(*grid),
                               pl.tile, rank, coldesc, mask_arg, desc_mode, occ_arg, n_cam, occ_stride, compact_form ? live : nullptr, clear_ptr, clear_floats, no_ranks ? 1 : 0);
        else if (four)
            hipLaunchKernelGGL((k_rank_columns4<8>), pgrid4, dim3(256), 0, s, geometry, frames * n_cam, D, H, W, to_params(*grid),
                               pl.tile, rank, coldesc, mask_arg, desc_mode, occ_arg, n_cam, occ_stride, compact_form ? live : nullptr, clear_ptr, clear_floats, no_ranks ? 1 : 0);
        else if (rows >= 28)
            hipLaunchKernelGGL((k_rank_columns<28>), pgrid, dim3(256), 0, s, geometry, frames * n_cam, D, H, W, to_params(*grid),
                               pl.tile, rank, coldesc, mask_arg, desc_mode, occ_arg, n_cam, occ_stride, compact_form ? live : nullptr, clear_ptr, clear_floats);
        else if (rows >= 14)
            hipLaunchKernelGGL((k_rank_columns<14>), pgrid, dim3(256), 0, s, geometry, frames * n_cam, D, H, W, to_params(*grid),
                               pl.tile, rank, coldesc, mask_arg, desc_mode, occ_arg, n_cam, occ_stride, compact_form ? live : nullptr, clear_ptr, clear_floats);
        else
            hipLaunchKernelGGL((k_rank_columns<8>), pgrid, dim3(256), 0, s, geometry, frames * n_cam, D, H, W, to_params(*grid),
                               pl.tile, rank, coldesc, mask_arg, desc_mode, occ_arg, n_cam, occ_stride, compact_form ? live : nullptr, clear_ptr, clear_floats);
    }
    rc = check_launch("rank_columns");
    if (rc) return rc;
    if (compact_form) {
        const int per_cu = cp_threads == 512 ? 2 : 1;
        const int tail = cp_tail, parts = cp_parts;
        const int tail_first = C * frames - tail;                        // (the first of the tail units)
        if (getenv("FIERY_POOL_VERBOSE"))
            fprintf(stderr, "voxel_pool compact: %d cells, %zu B LDS, %d threads, %d per CU, %d units + %d x %d parts\n", cp_cells,
                    cp_lds, cp_threads, per_cu, tail_first, tail, parts);
        const int n_items = tail_first + tail * parts;
        dim3 units(static_cast<unsigned>(persistent && n_items > cp_slots ? cp_slots : n_items));
        // (the counter lies in the cleared region in front of `occupied`)
        int* draw = (persistent && tail > 0 && !getenv("FIERY_POOL_NO_DRAW")) ? counter_block : nullptr;
        int* counters = counter_block;
        const PoolClean cl{reinterpret_cast<uint4*>(occ), static_cast<int>(occ_stride / 16), live, n_cam * D, frames};
        if (getenv("FIERY_POOL_NO_COUNTERS")) counters = nullptr;                   // tuning: no tickets, no in-kernel cleaning (callers must not pass WORKSPACE_CLEAN)
        int part_ranges = 0;
        if (const char* forced = getenv("FIERY_POOL_PART_RANGES")) part_ranges = atoi(forced);      // tuning / A-B runs
        int late_first = cp_slots / 2, late_prio = 0;
        if (const char* forced = getenv("FIERY_POOL_LATE_PRIO")) late_prio = atoi(forced);      // tuning / A-B runs
        long long* trace = nullptr;
        if (const char* t = getenv("FIERY_POOL_TRACE")) trace = reinterpret_cast<long long*>(strtoull(t, nullptr, 0));   // tuning
        const bool tuning = persistent || trace || late_prio != 0 || part_ranges != 0 || draw != nullptr;
        bool nt = true;                                                  // non-temporal row loads (see the kernel)
        if (const char* forced = getenv("FIERY_POOL_NT")) nt = atoi(forced) != 0;                // tuning / A-B runs
#define FIERY_POOL_COMPACT_LAUNCH(THREADS, WIDE, EXACT, NT)                                                              \
    do {                                                                                                                 \
        if (tuning) FIERY_POOL_COMPACT_LAUNCH_T(THREADS, WIDE, EXACT, NT, true);                                         \
        else FIERY_POOL_COMPACT_LAUNCH_T(THREADS, WIDE, EXACT, NT, false);                                               \
    } while (0)
#define FIERY_POOL_COMPACT_LAUNCH_T(THREADS, WIDE, EXACT, NT, TUNING)                                                    \
    do {                                                                                                                 \
        if (cp_lds > 65536 &&                                                                                            \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_pool_compact<THREADS, WIDE, EXACT, NT, TUNING>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(cp_lds)) != hipSuccess)     \
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot reserve %zu B of LDS", cp_lds);                              \
        hipLaunchKernelGGL((k_voxel_pool_compact<THREADS, WIDE, EXACT, NT, TUNING>), units, dim3(THREADS), cp_lds, s, x, st, rank, \
                           static_cast<const void*>(coldesc), occ, live, out, occupied, n_cam, D, H, W, C, pl.n_vox, pl.n_words, \
                           cp_cells, tail_first, parts, n_items, draw, trace, late_first, late_prio, counters, cl,       \
                           part_ranges);                                                                                \
    } while (0)
#define FIERY_POOL_COMPACT_ROWS(THREADS, WIDE)                            \
    do {                                                                 \
        if (H == 4 * kCompactRows && nt) FIERY_POOL_COMPACT_LAUNCH(THREADS, WIDE, true, true);  \
        else if (H == 4 * kCompactRows) FIERY_POOL_COMPACT_LAUNCH(THREADS, WIDE, true, false);  \
        else FIERY_POOL_COMPACT_LAUNCH(THREADS, WIDE, false, false);      \
    } while (0)
        if (!wide_records) {
            if (cp_threads == 512) FIERY_POOL_COMPACT_ROWS(512, false);
            else FIERY_POOL_COMPACT_ROWS(1024, false);
        } else {
            if (cp_threads == 512) FIERY_POOL_COMPACT_ROWS(512, true);
            else FIERY_POOL_COMPACT_ROWS(1024, true);
        }
#undef FIERY_POOL_COMPACT_ROWS
#undef FIERY_POOL_COMPACT_LAUNCH
#undef FIERY_POOL_COMPACT_LAUNCH_T
        return check_launch("voxel_pool (compact)");
    }
    if (plane_form) {
        int batch = 16;
        if (H % 16 != 0 && H % 7 == 0) batch = 7;
        if (const char* forced = getenv("FIERY_POOL_BATCH")) {                                  // tuning / tests
            const int b = atoi(forced);
            if (b == 7 || b == 8 || b == 16) batch = b;
        }
        // what the plane leaves of the CU's LDS holds the queue of many-run quads
        const long long spare = (163840 - static_cast<long long>(pl.n_vox) * 4) / 4 - 1;
        int queue_cap = static_cast<int>(spare < n_quads ? (spare < 0 ? 0 : spare) : n_quads);
        if (queue_cap > 1024) queue_cap = 1024;       // many-run quads are rare; a small plane should leave LDS for more workgroups
        if (const char* forced = getenv("FIERY_POOL_QUEUE_CAP")) queue_cap = min(queue_cap, max(0, atoi(forced)));   // tests
        const size_t lds = (static_cast<size_t>(pl.n_vox) + 1 + queue_cap) * 4;
        const int wg_threads = lds <= 40960 ? 256 : (lds <= 81920 ? 512 : 1024);
        auto fastdiv = [](int d) { return FastDiv{((1ull << 40) + d - 1) / static_cast<unsigned long long>(d)}; };
        const int2* cd = reinterpret_cast<const int2*>(coldesc);
        // one workgroup per CU (the plane takes more than half of the LDS): cut the units of the last, partly filled
        // round into parts (see the kernel); 576 planes on 256 CUs -> the last 64 planes become 256 workgroups
        const int n_units = C * frames;
        int tail = 0, parts = 1;
        if (lds > 81920) {
            int dev = 0, n_cu = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
                n_cu = 256;
            tail = n_units % n_cu;
            parts = tail > 0 ? n_cu / tail : 1;
        }
        if (const char* forced = getenv("FIERY_POOL_TAIL_PARTS")) {                           // tuning / tests
            parts = atoi(forced);
            if (parts > 1 && tail == 0) tail = n_units < 3 ? n_units : 3;
        }
        if (parts > 8) parts = 8;
        if (parts < 2) {
            parts = 1;
            tail = 0;
        }
        const int tail_first = n_units - tail;
        if (tail > 0 && hipMemsetAsync(out + static_cast<long long>(tail_first) * pl.n_vox, 0,
                                       static_cast<size_t>(tail) * pl.n_vox * sizeof(float), s) != hipSuccess)
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot clear the tail planes");
        dim3 units(static_cast<unsigned>(tail_first + tail * parts));
#define FIERY_POOL_PLANE_LAUNCH(BATCH)                                                                                   \
    do {                                                                                                                 \
        if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_pool_plane<BATCH>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) \
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot reserve %zu B of LDS", lds);                                 \
        hipLaunchKernelGGL((k_voxel_pool_plane<BATCH>), units, dim3(wg_threads), lds, s, x, st, rank, cd, out, n_cam, D, H, W, \
                           C, pl.n_vox, queue_cap, fastdiv(W / 4), fastdiv(D), tail_first, parts);                       \
    } while (0)
        if (batch == 7) FIERY_POOL_PLANE_LAUNCH(7);
        else if (batch == 8) FIERY_POOL_PLANE_LAUNCH(8);
        else FIERY_POOL_PLANE_LAUNCH(16);
#undef FIERY_POOL_PLANE_LAUNCH
        return check_launch("voxel_pool (plane)");
    }
    hipLaunchKernelGGL(k_build_tile_lists, dim3(pl.n_tiles, frames), dim3(1024), 0, s, colmask, n_cam, D, W,
                       quads ? 4 : 1, pl.n_tiles, lists, counts);
    rc = check_launch("build_tile_lists");
    if (rc) return rc;
    const long long n_units = static_cast<long long>(C) * frames * pl.n_tiles;
    FIERY_REQUIRE(n_units < (1ll << 31), "voxel_pool: too many (tile, channel, frame) units");
    dim3 gridDim3(static_cast<unsigned>(n_units));
    // rows of a work-item in flight at once: a count that divides H spares the row predicates and zero fills of a
    // ragged last batch (H = 28: 14); the fused form holds two operands per row, hence its shorter batches
    int batch = fused ? 8 : 16;
    if (H % batch != 0) {
        if (H % 14 == 0 || (fused && H % 7 == 0)) batch = 7;       // H = 28: four batches of 7 (measured best)
    }
    if (const char* forced = getenv("FIERY_POOL_BATCH")) {                                      // tuning / tests
        const int b = atoi(forced);
        if (b == 4 || b == 7 || b == 8 || (!fused && (b == 14 || b == 16))) batch = b;
    }
    int balanced = 1;                           // unit order, see k_voxel_pool
    if (const char* forced = getenv("FIERY_POOL_ORDER")) balanced = atoi(forced) != 0;         // tuning / A-B runs
#define FIERY_POOL_LAUNCH(VEC, BATCH, FUSED, FIXED)                                                                      \
    do {                                                                                                                 \
        if (pl.lds > 65536 &&                                                                                            \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_pool<VEC, BATCH, FUSED, FIXED>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pl.lds)) != hipSuccess)      \
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot reserve %zu B of LDS", pl.lds);                              \
        hipLaunchKernelGGL((k_voxel_pool<VEC, BATCH, FUSED, FIXED>), gridDim3, dim3(threads), pl.lds, s, x, st, depth, feat, \
                           rank, coldesc, lists, counts, out, n_cam, D, H, W, C, pl.n_vox, pl.tile, pl.n_tiles, frames, \
                           balanced);                                                                                    \
    } while (0)
#define FIERY_POOL_DISPATCH(VEC, BATCH)                          \
    do {                                                         \
        if (fused && fixed) FIERY_POOL_LAUNCH(VEC, BATCH, true, true);        \
        else if (fused) FIERY_POOL_LAUNCH(VEC, BATCH, true, false);           \
        else if (fixed) FIERY_POOL_LAUNCH(VEC, BATCH, false, true);           \
        else FIERY_POOL_LAUNCH(VEC, BATCH, false, false);                     \
    } while (0)
#define FIERY_POOL_LAUNCH_PIPE(BATCH, FIXED)                                                                             \
    do {                                                                                                                 \
        if (pl.lds > 65536 &&                                                                                            \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_pool<4, BATCH, false, FIXED, false, true>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pl.lds)) != hipSuccess)      \
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot reserve %zu B of LDS", pl.lds);                              \
        hipLaunchKernelGGL((k_voxel_pool<4, BATCH, false, FIXED, false, true>), gridDim3, dim3(threads), pl.lds, s, x, st, \
                           depth, feat, rank, coldesc, lists, counts, out, n_cam, D, H, W, C, pl.n_vox, pl.tile,         \
                           pl.n_tiles, frames, balanced);                                                                \
    } while (0)
    // The software-pipelined walk (two alternating row buffers) needs an even number of whole batches per column.
    // Opt-in (FIERY_POOL_PIPE=1): on MI355X keeping twice the rows in flight made the kernel slower, not faster (350 vs
    // 330 us at batch 7, 505 vs 362 us at batch 14: the loads already queue behind the memory system).
    const char* pipe_env = getenv("FIERY_POOL_PIPE");
    bool pipe = pipe_env && atoi(pipe_env) != 0 && quads && !fused && (batch == 7 || batch == 14) && H % (2 * batch) == 0;
    if (pipe) {
        // its buffer loads address a (frame, channel) plane with 31-bit byte offsets
        const long long reach = (n_cam - 1) * st.n + (D - 1) * st.d + (H - 1) * st.h + (W - 1) * st.w + 4;
        pipe = st.n >= 0 && st.d >= 0 && st.h >= 0 && reach < (1ll << 29);
    }
    if (getenv("FIERY_POOL_PROBE")) pipe = false;
    if (pipe) {
        if (batch == 7) {
            if (fixed) FIERY_POOL_LAUNCH_PIPE(7, true);
            else FIERY_POOL_LAUNCH_PIPE(7, false);
        } else {
            if (fixed) FIERY_POOL_LAUNCH_PIPE(14, true);
            else FIERY_POOL_LAUNCH_PIPE(14, false);
        }
    } else if (const char* probe = getenv("FIERY_POOL_PROBE")) {
        unsigned long long* ptr = reinterpret_cast<unsigned long long*>(strtoull(probe, nullptr, 0));
        FIERY_REQUIRE(quads && !fused && !fixed, "voxel_pool: the probe variant exists for the 16-byte fp32 path only");
        if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_pool_probe), &ptr, sizeof(ptr), 0, hipMemcpyHostToDevice, s) != hipSuccess)
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot set up the probe");
#define FIERY_POOL_PROBE_LAUNCH(BATCH)                                                                                   \
    do {                                                                                                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_pool<4, BATCH, false, false, true>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pl.lds)) != hipSuccess)      \
            return fail(FIERY_ELAUNCH, "voxel_pool: cannot set up the probe");                                           \
        hipLaunchKernelGGL((k_voxel_pool<4, BATCH, false, false, true>), gridDim3, dim3(threads), pl.lds, s, x, st, depth, \
                           feat, rank, coldesc, lists, counts, out, n_cam, D, H, W, C, pl.n_vox, pl.tile, pl.n_tiles,    \
                           frames, balanced);                                                                            \
    } while (0)
        if (batch == 14) FIERY_POOL_PROBE_LAUNCH(14);
        else FIERY_POOL_PROBE_LAUNCH(16);
#undef FIERY_POOL_PROBE_LAUNCH
    } else if (quads) {
        if (batch == 4) FIERY_POOL_DISPATCH(4, 4);
        else if (batch == 7) FIERY_POOL_DISPATCH(4, 7);
        else if (batch == 8) FIERY_POOL_DISPATCH(4, 8);
        else if (batch == 14) FIERY_POOL_DISPATCH(4, 14);
        else FIERY_POOL_DISPATCH(4, 16);
    } else {
        if (batch <= 8) FIERY_POOL_DISPATCH(1, 8);
        else FIERY_POOL_DISPATCH(1, 16);
    }
#undef FIERY_POOL_DISPATCH
#undef FIERY_POOL_LAUNCH_PIPE
#undef FIERY_POOL_LAUNCH
    return check_launch("voxel_pool");
}

}  // namespace

extern "C" int fiery_voxel_pool_fwd(const float* x, const int64_t* x_strides, const float* geometry, int frames,
                                    int n_cameras, int D, int H, int W, int C, const fiery_bev_grid* grid, float* out,
                                    void* workspace, size_t workspace_bytes, int tile_voxels, uint32_t flags,
                                    fiery_stream_t stream) {
    return pool_common(false, x, x_strides, nullptr, nullptr, geometry, frames, n_cameras, D, H, W, C, grid, out,
                       workspace, workspace_bytes, tile_voxels, flags, stream);
}

extern "C" int fiery_lift_splat_fwd(const float* depth_prob, const float* features, const float* geometry, int frames,
                                    int n_cameras, int D, int H, int W, int C, const fiery_bev_grid* grid, float* out,
                                    void* workspace, size_t workspace_bytes, int tile_voxels, uint32_t flags,
                                    fiery_stream_t stream) {
    return pool_common(true, nullptr, nullptr, depth_prob, features, geometry, frames, n_cameras, D, H, W, C, grid, out,
                       workspace, workspace_bytes, tile_voxels, flags, stream);
}

// ------------------------------------------------------------------------------------------------
// backward of the pooling (training)
// ------------------------------------------------------------------------------------------------
namespace fiery {
namespace {

// d(out)/d(x): every in-grid point receives the gradient of the voxel it was added to, every other point 0 -
// what `VoxelsSumming.backward` (fiery/utils/geometry.py:304-314: grad_out[cumsum(keep) - keep]) amounts to once
// autograd has undone the sort, the mask and the reshape around it (fiery.py:233-261).  A pure copy, so bit-exact.
// HBM-bound on the write of grad_x: a thread owns kVec adjacent points (one 16-byte rank load), keeps their ranks in
// registers and walks kChan channels - eight channels' gathers (L2-resident: a channel plane is n_vox floats, and
// the 28 rows of a column name the same voxel) are in flight before their eight 16-byte streaming stores.
// grid (ceil(D*H*W / kVec / 256), ceil(C / kChan), frames * n_cam)
template <int kVec, int kChan>
__global__ __launch_bounds__(256) void k_voxel_pool_bwd(const float* __restrict__ g, const int* __restrict__ rank,
                                                         float* __restrict__ gx, PoolStrides gs, int n_cam, int D, int H,
                                                         int W, int C, int n_vox) {
    const int Wg = W / kVec;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= D * H * Wg) return;
    const int w = (item % Wg) * kVec;
    const int h = (item / Wg) % H;
    const int d = item / (Wg * H);
    const int fc = blockIdx.z;
    const int f = fc / n_cam, cam = fc - f * n_cam;
    const int* rp = rank + (static_cast<long long>(fc) * D + d) * H * W + h * W + w;
    int r[kVec];
    if (kVec == 4) {
        const int4 t = *reinterpret_cast<const int4*>(rp);
        r[0] = t.x;  r[1 % kVec] = t.y;  r[2 % kVec] = t.z;  r[3 % kVec] = t.w;
    } else {
        r[0] = rp[0];
    }
    const int c0 = blockIdx.y * kChan;
    const int c1 = min(c0 + kChan, C);
    const float* gp = g + (static_cast<long long>(f) * C + c0) * n_vox;
    float* op = gx + f * gs.f + cam * gs.n + d * gs.d + h * gs.h + w * gs.w + c0 * gs.c;
    constexpr int kUnroll = 8;
    for (int c = c0; c < c1; c += kUnroll) {
        float v[kUnroll][kVec];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j)
#pragma unroll
            for (int k = 0; k < kVec; ++k)
                v[j][k] = (c + j < c1 && r[k] >= 0) ? gp[static_cast<long long>(j) * n_vox + r[k]] : 0.f;
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            if (c + j >= c1) break;
            if (kVec == 4) {
                vf4 t = {v[j][0], v[j][1 % kVec], v[j][2 % kVec], v[j][3 % kVec]};
                __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(op + j * gs.c));     // written once, read by another kernel
            } else {
                op[j * gs.c] = v[j][0];
            }
        }
        gp += static_cast<long long>(kUnroll) * n_vox;
        op += kUnroll * gs.c;
    }
}

// Backward of the fused lift (x) splat, out = sum_points depth * feat:
//   grad_depth[f][n][d][h][w] = sum_c feat[f][n][c][h][w] * g[f][c][rank(f,n,d,h,w)]
//   grad_feat [f][n][c][h][w] = sum_d depth[f][n][d][h][w] * g[f][c][rank(f,n,d,h,w)]
// (autograd through fiery/models/encoder.py:99-100 and the pooling).  The (n, C, D, H, W) gradient of the outer
// product - 372 MB per sample - never exists.  Both sums need, for every point, the C gradients of its voxel: in the
// channel-planar g those are C cache lines, which made a first version gather-bound (2.0 ms per 9 frames).  So g is
// first transposed to voxel-major gT[f][voxel][C] (workspace), where a point's gradients are C contiguous floats, and
// one thread per *pixel* (f, n, h, w) keeps its feature vector and its grad_feat accumulators in registers and walks
// the D points of its ray: 16-byte loads of gT rows (neighbouring pixels share them), a dot product for grad_depth, an
// axpy for grad_feat.  Channels in chunks of kChunk (registers); sums run in a fixed order: reproducible.
// grid (ceil(H*W / 64), 1, frames * n_cam), 64 threads
template <int kChunk>
__global__ __launch_bounds__(64) void k_lift_splat_bwd(const float* __restrict__ gT, int ld, const int* __restrict__ rank,
                                                        const float* __restrict__ depth, const float* __restrict__ feat,
                                                        float* __restrict__ gdepth, float* __restrict__ gfeat, int D, int HW,
                                                        int C, int n_cam, int n_vox) {
    const int hw = blockIdx.x * blockDim.x + threadIdx.x;
    if (hw >= HW) return;
    const int fc = blockIdx.z;
    const int f = fc / n_cam;
    const int* rp = rank + static_cast<long long>(fc) * D * HW + hw;
    const float* dp = depth + static_cast<long long>(fc) * D * HW + hw;
    float* gdp = gdepth ? gdepth + static_cast<long long>(fc) * D * HW + hw : nullptr;
    const float* gbase = gT + static_cast<long long>(f) * n_vox * ld;
    for (int c0 = 0; c0 < C; c0 += kChunk) {
        float ft[kChunk], acc[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            ft[j] = (c0 + j < C) ? feat[(static_cast<long long>(fc) * C + c0 + j) * HW + hw] : 0.f;
            acc[j] = 0.f;
        }
        for (int d = 0; d < D; ++d) {
            const int r = rp[static_cast<long long>(d) * HW];
            const float pd = dp[static_cast<long long>(d) * HW];
            float dot = 0.f;
            if (r >= 0) {
                const float4* row = reinterpret_cast<const float4*>(gbase + static_cast<long long>(r) * ld + c0);
                float g[kChunk];
#pragma unroll
                for (int j = 0; j < kChunk; j += 4) {
                    // the padding columns of gT (C..ld) are zero, so a chunk that overhangs C reads zeros or stops
                    const float4 t = (c0 + j < ld) ? row[j / 4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    g[j] = t.x;  g[j + 1] = t.y;  g[j + 2] = t.z;  g[j + 3] = t.w;
                }
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    dot += ft[j] * g[j];
                    acc[j] += pd * g[j];
                }
            }
            if (gdp) {
                // later chunks add to what this same thread stored for the earlier ones
                float* o = gdp + static_cast<long long>(d) * HW;
                *o = c0 == 0 ? dot : *o + dot;
            }
        }
        if (gfeat) {
#pragma unroll
            for (int j = 0; j < kChunk; ++j)
                if (c0 + j < C) gfeat[(static_cast<long long>(fc) * C + c0 + j) * HW + hw] = acc[j];
        }
    }
}

// gT[f][v][0..ld) = g[f][0..C)[v], columns C..ld zero.  64 voxels per workgroup through LDS: unit-stride on both sides.
__global__ __launch_bounds__(256) void k_voxel_major(const float* __restrict__ g, int C, int n_vox, int ld, float* __restrict__ gT) {
    HIP_DYNAMIC_SHARED(float, vm_tile)   // [64][C + 1]
    const int f = blockIdx.y;
    const int v0 = blockIdx.x * 64;
    const int nv = min(64, n_vox - v0);
    const int row = C + 1;
    const float* src = g + static_cast<long long>(f) * C * n_vox + v0;
    for (int i = threadIdx.x; i < C * 64; i += blockDim.x) {
        const int c = i >> 6, v = i & 63;
        if (v < nv) vm_tile[v * row + c] = src[static_cast<long long>(c) * n_vox + v];
    }
    __syncthreads();
    float* dst = gT + (static_cast<long long>(f) * n_vox + v0) * ld;
    for (int i = threadIdx.x; i < nv * ld; i += blockDim.x) {
        const int v = i / ld, c = i - v * ld;
        dst[i] = c < C ? vm_tile[v * row + c] : 0.f;
    }
}

}  // namespace
}  // namespace fiery

extern "C" int fiery_voxel_pool_bwd(const float* grad_out, const int32_t* rank, int frames, int n_cameras, int D, int H,
                                    int W, int C, int n_voxels, float* grad_x, const int64_t* gx_strides,
                                    fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && rank && grad_x && gx_strides, "voxel_pool_bwd: null pointer");
    FIERY_REQUIRE(frames > 0 && n_cameras > 0 && D > 0 && H > 0 && W > 0 && C > 0 && n_voxels > 0, "voxel_pool_bwd: bad shape");
    FIERY_REQUIRE(static_cast<long long>(frames) * n_cameras < 65536, "voxel_pool_bwd: frames * cameras >= 65536");
    const PoolStrides st{gx_strides[0], gx_strides[1], gx_strides[2], gx_strides[3], gx_strides[4], gx_strides[5]};
    // 16-byte path: four adjacent points of a row are contiguous and aligned in both the rank array and grad_x
    bool quads = W % 4 == 0 && aligned16(rank) && aligned16(grad_x) && st.w == 1 && st.f % 4 == 0 && st.n % 4 == 0 &&
                 st.d % 4 == 0 && st.h % 4 == 0 && st.c % 4 == 0;
    if (const char* forced = getenv("FIERY_POOL_VEC")) quads = quads && atoi(forced) == 4;      // tuning / tests
    constexpr int kChan = 16;
    const int vec = quads ? 4 : 1;
    dim3 grid3(ceil_div(static_cast<long long>(D) * H * (W / vec), 256), ceil_div(C, kChan), frames * n_cameras);
    if (quads)
        hipLaunchKernelGGL((fiery::k_voxel_pool_bwd<4, kChan>), grid3, dim3(256), 0, as_stream(stream), grad_out, rank, grad_x,
                           st, n_cameras, D, H, W, C, n_voxels);
    else
        hipLaunchKernelGGL((fiery::k_voxel_pool_bwd<1, kChan>), grid3, dim3(256), 0, as_stream(stream), grad_out, rank, grad_x,
                           st, n_cameras, D, H, W, C, n_voxels);
    return check_launch("voxel_pool_bwd");
}

extern "C" size_t fiery_lift_splat_bwd_workspace_bytes(int frames, int C, int n_voxels) {
    if (frames <= 0 || C <= 0 || n_voxels <= 0) return 0;
    return static_cast<size_t>(frames) * n_voxels * ((C + 3) / 4 * 4) * sizeof(float);
}

extern "C" int fiery_lift_splat_bwd(const float* grad_out, const int32_t* rank, const float* depth_prob,
                                    const float* features, int frames, int n_cameras, int D, int H, int W, int C,
                                    int n_voxels, float* grad_depth, float* grad_features, void* workspace,
                                    size_t workspace_bytes, fiery_stream_t stream) {
    FIERY_REQUIRE(grad_out && rank && depth_prob && features && workspace, "lift_splat_bwd: null pointer");
    FIERY_REQUIRE(grad_depth || grad_features, "lift_splat_bwd: nothing to compute");
    FIERY_REQUIRE(frames > 0 && n_cameras > 0 && D > 0 && H > 0 && W > 0 && C > 0 && n_voxels > 0, "lift_splat_bwd: bad shape");
    FIERY_REQUIRE(static_cast<long long>(frames) * n_cameras < 65536 && frames < 65536, "lift_splat_bwd: frames * cameras >= 65536");
    FIERY_REQUIRE(aligned16(workspace), "lift_splat_bwd: the workspace must be 16-byte aligned");
    FIERY_REQUIRE(static_cast<size_t>(64) * (C + 1) * sizeof(float) <= 160 * 1024, "lift_splat_bwd: too many channels");
    const size_t need = fiery_lift_splat_bwd_workspace_bytes(frames, C, n_voxels);
    if (workspace_bytes < need) return fail(FIERY_ENOMEM, "lift_splat_bwd: workspace %zu B < required %zu B", workspace_bytes, need);
    const int HW = H * W;
    const int ld = (C + 3) / 4 * 4;
    float* gT = static_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
    const size_t lds = static_cast<size_t>(64) * (C + 1) * sizeof(float);
    if (lds > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(&fiery::k_voxel_major),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
        return fail(FIERY_ELAUNCH, "lift_splat_bwd: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(fiery::k_voxel_major, dim3(ceil_div(n_voxels, 64), frames), dim3(256), lds, s, grad_out, C, n_voxels, ld, gT);
    int rc = check_launch("lift_splat_bwd (transpose)");
    if (rc) return rc;
    hipLaunchKernelGGL((fiery::k_lift_splat_bwd<32>), dim3(ceil_div(HW, 64), 1, frames * n_cameras), dim3(64), 0, s, gT, ld, rank,
                       depth_prob, features, grad_depth, grad_features, D, HW, C, n_cameras, n_voxels);
    return check_launch("lift_splat_bwd");
}

namespace fiery {
namespace {
__global__ void k_depth_softmax(const float* __restrict__ logits, int n, int D, int HW, float* __restrict__ prob) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(n) * HW) return;
    const int img = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(img) * HW);
    const float* in = logits + static_cast<long long>(img) * D * HW + p;
    float* o = prob + static_cast<long long>(img) * D * HW + p;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, in[static_cast<long long>(d) * HW]);
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += expf(in[static_cast<long long>(d) * HW] - m);
    for (int d = 0; d < D; ++d) o[static_cast<long long>(d) * HW] = expf(in[static_cast<long long>(d) * HW] - m) / s;
}
// softmax backward over the depth axis: grad_logits = p * (grad_p - sum_d p * grad_p)
__global__ void k_depth_softmax_bwd(const float* __restrict__ prob, const float* __restrict__ gprob, int n, int D, int HW,
                                    float* __restrict__ glogits) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(n) * HW) return;
    const int img = static_cast<int>(i / HW);
    const long long base = static_cast<long long>(img) * D * HW + (i - static_cast<long long>(img) * HW);
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot += prob[base + static_cast<long long>(d) * HW] * gprob[base + static_cast<long long>(d) * HW];
    for (int d = 0; d < D; ++d) {
        const long long j = base + static_cast<long long>(d) * HW;
        glogits[j] = prob[j] * (gprob[j] - dot);
    }
}
}  // namespace
}  // namespace fiery

extern "C" int fiery_depth_softmax_bwd(const float* prob, const float* grad_prob, int n, int D, int HW, float* grad_logits,
                                       fiery_stream_t stream) {
    FIERY_REQUIRE(prob && grad_prob && grad_logits && n > 0 && D > 0 && HW > 0, "depth_softmax_bwd: bad argument");
    hipLaunchKernelGGL(fiery::k_depth_softmax_bwd, dim3(ceil_div(static_cast<long long>(n) * HW, 256)), dim3(256), 0,
                       as_stream(stream), prob, grad_prob, n, D, HW, grad_logits);
    return check_launch("depth_softmax_bwd");
}

extern "C" int fiery_depth_softmax(const float* logits, int n, int D, int HW, float* prob, fiery_stream_t stream) {
    FIERY_REQUIRE(logits && prob && n > 0 && D > 0 && HW > 0, "depth_softmax: bad argument");
    hipLaunchKernelGGL(fiery::k_depth_softmax, dim3(ceil_div(static_cast<long long>(n) * HW, 256)), dim3(256), 0,
                       as_stream(stream), logits, n, D, HW, prob);
    return check_launch("depth_softmax");
}
