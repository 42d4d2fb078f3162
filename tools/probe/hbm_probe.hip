// Tuning aid, not part of the product library: what a plain streaming read reaches on this box, for a given
// (workgroups, threads, loads in flight per lane).  Built by tools/microbench.py into tools/probe/libhbm_probe.so.
#include <hip/hip_runtime.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
    const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int kUnroll, bool kNonTemporal>
__global__ void k_read(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (kUnroll - 1) * stride < n4; i += kUnroll * stride) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
            v[u] = kNonTemporal ? nt_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) acc += x[i].x;
    if (acc == 1.2345e-30f) sink[0] = acc;      // keeps the loads alive
}

// each workgroup streams its own contiguous slab (the pooling kernel's pattern) instead of a grid-stride sweep
template <int kUnroll>
__global__ void k_read_slabs(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    float acc = 0.f;
    long long i = lo + threadIdx.x;
    for (; i + (kUnroll - 1) * blockDim.x < hi; i += kUnroll * blockDim.x) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = x[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < hi; i += blockDim.x) acc += x[i].x;
    if (acc == 1.2345e-30f) sink[0] = acc;
}

// the pooling kernel's pattern: the tensor is a sequence of (28 rows x 15 quads) slices; a lane owns one quad column
// of a slice and reads its 28 rows (240 B apart), kBatch rows in flight; neighbouring lanes own neighbouring columns
template <int kBatch>
__global__ void k_read_columns(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    const long long n_items = n4 / 28;
    const long long per = (n_items + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < n_items ? lo + per : n_items;
    float acc = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float4* p = x + (i / 15) * 420 + (i % 15);
        for (int h0 = 0; h0 < 28; h0 += kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v[j] = h0 + j < 28 ? p[(h0 + j) * 15] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

// the same columns with their rows dealt out to R neighbouring lane groups: lane group r of a slice reads rows r, r + R, ...
// so that one wavefront instruction covers R adjacent 240-byte rows (R * 240 contiguous bytes per slice) instead of one
template <int R>
__global__ void k_read_columns_interleaved(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    const long long n_items = n4 / 28 * R;
    const long long per = (n_items + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < n_items ? lo + per : n_items;
    float acc = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const long long slice = i / (15 * R);
        const int within = static_cast<int>(i - slice * (15 * R));
        const int r = within / 15, col = within - r * 15;
        const float4* p = x + slice * 420 + col + r * 15;
        float4 v[28 / R];
#pragma unroll
        for (int j = 0; j < 28 / R; ++j) v[j] = p[j * R * 15];
#pragma unroll
        for (int j = 0; j < 28 / R; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

// the whole-plane pooling kernel's reads without its descriptors: workgroup (channel, frame) walks the 6 x 48 slices of
// its channel (6 separate 322 KB regions, 20.6 MB apart), a lane owns a quad column, kBatch rows in flight
template <int kBatch, bool kNT>
__global__ __launch_bounds__(1024) void k_read_planes(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    extern __shared__ float lds[];
    const int c = blockIdx.x % 64, f = blockIdx.x / 64;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 6 * 48 * 15; i += blockDim.x) {
        const int cam = i / (48 * 15), rem = i - cam * 48 * 15, d = rem / 15, col = rem - d * 15;
        const float4* p = x + ((((long long)f * 6 + cam) * 64 + c) * 48 + d) * 420 + col;
        for (int h0 = 0; h0 < 28; h0 += kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v[j] = kNT ? nt_load(p + (h0 + j) * 15) : p[(h0 + j) * 15];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    }
    if (acc == 1.2345e-30f) { sink[0] = acc; lds[threadIdx.x] = acc; }
}

extern "C" int probe_read(const void* x, long long n_bytes, int blocks, int threads, int unroll, int mode, void* sink,
                          void* stream) {
    const float4* p = static_cast<const float4*>(x);
    const long long n4 = n_bytes / 16;
    float* s = static_cast<float*>(sink);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GO(K, U) hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, st, p, n4, s)
    if (mode == 0) {
        if (unroll == 1) GO((k_read<1, false>), 1); else if (unroll == 4) GO((k_read<4, false>), 4);
        else if (unroll == 8) GO((k_read<8, false>), 8); else GO((k_read<16, false>), 16);
    } else if (mode == 1) {
        if (unroll == 1) GO((k_read<1, true>), 1); else if (unroll == 4) GO((k_read<4, true>), 4);
        else if (unroll == 8) GO((k_read<8, true>), 8); else GO((k_read<16, true>), 16);
    } else if (mode == 5 || mode == 6) {
        const int lds = 160000;
        if (mode == 5) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read_planes<7, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((k_read_planes<7, true>), dim3(blocks), dim3(threads), lds, st, p, n4, s);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read_planes<7, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((k_read_planes<7, false>), dim3(blocks), dim3(threads), lds, st, p, n4, s);
        }
    } else if (mode == 4) {
        if (unroll == 2) GO((k_read_columns_interleaved<2>), 2); else if (unroll == 4) GO((k_read_columns_interleaved<4>), 4);
        else GO((k_read_columns_interleaved<1>), 1);
    } else if (mode == 3) {
        if (unroll <= 8) GO((k_read_columns<8>), 8); else if (unroll <= 16) GO((k_read_columns<16>), 16);
        else GO((k_read_columns<28>), 28);
    } else {
        if (unroll == 1) GO((k_read_slabs<1>), 1); else if (unroll == 4) GO((k_read_slabs<4>), 4);
        else if (unroll == 8) GO((k_read_slabs<8>), 8); else GO((k_read_slabs<16>), 16);
    }
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- round 2: the (channel, frame) unit with a SMALLER LDS footprint (compact plane of occupied voxels only), so that
// two or three workgroups share a CU; R = 1: a lane owns a quad column and reads kBatch rows at a time; R = 4: the 28 rows
// of a slice are dealt to four 15-lane groups (lanes 16 g + col, lane 15 of each group idle), 7 loads per lane and item,
// every wavefront instruction covers 960 contiguous bytes
template <int kBatch, int kThreads, int R>
__global__ __launch_bounds__(kThreads) void k_read_planes2(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    extern __shared__ float lds[];
    const int c = blockIdx.x % 64, f = blockIdx.x / 64;
    float acc = 0.f;
    if (R == 1) {
        for (int i = threadIdx.x; i < 6 * 48 * 15; i += kThreads) {
            const int cam = i / (48 * 15), rem = i - cam * 48 * 15, d = rem / 15, col = rem - d * 15;
            const float4* p = x + ((((long long)f * 6 + cam) * 64 + c) * 48 + d) * 420 + col;
            for (int h0 = 0; h0 < 28; h0 += kBatch) {
                float4 v[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) v[j] = p[(h0 + j) * 15];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
            }
        }
    } else {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = kThreads / 64;
        const int g = lane >> 4, col = lane & 15;
        for (int s = wave; s < 6 * 48; s += n_waves) {            // one (camera, depth) slice per wavefront and step
            const int cam = s / 48, d = s - cam * 48;
            const float4* p = x + ((((long long)f * 6 + cam) * 64 + c) * 48 + d) * 420 + g * 15 + col;
            float4 v[7];
            if (col < 15) {
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = p[j * 60];
#pragma unroll
                for (int j = 0; j < 7; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
            }
        }
    }
    if (acc == 1.2345e-30f) { sink[0] = acc; lds[threadIdx.x] = acc; }
}

extern "C" int probe_planes2(const void* x, long long n_bytes, int blocks, int threads, int batch, int R, int lds_bytes,
                             void* sink, void* stream) {
    const float4* p = static_cast<const float4*>(x);
    const long long n4 = n_bytes / 16;
    float* s = static_cast<float*>(sink);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GO2(B, T, RR)                                                                                                        \
    do {                                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read_planes2<B, T, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
        hipLaunchKernelGGL((k_read_planes2<B, T, RR>), dim3(blocks), dim3(T), lds_bytes, st, p, n4, s);                       \
    } while (0)
#define GOT(T)                                                        \
    do {                                                              \
        if (R == 4) GO2(7, T, 4);                                     \
        else if (batch == 14) GO2(14, T, 1);                          \
        else if (batch == 28) GO2(28, T, 1);                          \
        else GO2(7, T, 1);                                            \
    } while (0)
    if (threads == 256) GOT(256);
    else if (threads == 384) GOT(384);
    else if (threads == 512) GOT(512);
    else if (threads == 768) GOT(768);
    else GOT(1024);
#undef GOT
#undef GO2
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- round 2, second probe: the row-dealt slice pattern with more slices in flight per wavefront (kSlices x 7 loads per
// lane before anything is consumed), plain or non-temporal, and a rolling variant (kRolling: a slot is re-requested for
// the next slice as soon as its row is consumed, so 7 loads per lane are in flight at ALL times)
template <int kThreads, int kSlices, bool kNT, bool kRolling>
__global__ __launch_bounds__(kThreads) void k_read_planes3(const float4* __restrict__ x, long long n4, float* __restrict__ sink) {
    extern __shared__ float lds[];
    const int c = blockIdx.x % 64, f = blockIdx.x / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int n_waves = kThreads / 64;
    const int g = lane >> 4, col = lane & 15;
    float acc = 0.f;
    auto slice_ptr = [&](int s) {
        const int cam = s / 48, d = s - cam * 48;
        return x + ((((long long)f * 6 + cam) * 64 + c) * 48 + d) * 420 + g * 15 + col;
    };
    auto ld = [&](const float4* p) { return kNT ? nt_load(p) : *p; };
    if (col < 15) {
        if (kRolling) {
            float4 v[7];
            int s = wave;
            const float4* p = slice_ptr(s);
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = ld(p + j * 60);
            for (; s < 6 * 48; s += n_waves) {
                const bool more = s + n_waves < 6 * 48;
                const float4* pn = slice_ptr(more ? s + n_waves : s);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float4 r = v[j];
                    if (more) v[j] = ld(pn + j * 60);
                    acc += r.x + r.y + r.z + r.w;
                }
            }
        } else {
            for (int s0 = wave * kSlices; s0 < 6 * 48; s0 += n_waves * kSlices) {
                float4 v[kSlices][7];
#pragma unroll
                for (int u = 0; u < kSlices; ++u) {
                    const float4* p = slice_ptr(s0 + u < 6 * 48 ? s0 + u : s0);
#pragma unroll
                    for (int j = 0; j < 7; ++j) v[u][j] = ld(p + j * 60);
                }
#pragma unroll
                for (int u = 0; u < kSlices; ++u)
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc += v[u][j].x + v[u][j].y + v[u][j].z + v[u][j].w;
            }
        }
    }
    if (acc == 1.2345e-30f) { sink[0] = acc; lds[threadIdx.x] = acc; }
}

extern "C" int probe_planes3(const void* x, long long n_bytes, int blocks, int threads, int slices, int nt, int rolling,
                             int lds_bytes, void* sink, void* stream) {
    const float4* p = static_cast<const float4*>(x);
    const long long n4 = n_bytes / 16;
    float* s = static_cast<float*>(sink);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GO3(T, S, N, R)                                                                                                      \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read_planes3<T, S, N, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
        hipLaunchKernelGGL((k_read_planes3<T, S, N, R>), dim3(blocks), dim3(T), lds_bytes, st, p, n4, s);                     \
    } while (0)
#define GO3T(T)                                                                     \
    do {                                                                            \
        if (rolling) { if (nt) GO3(T, 1, true, true); else GO3(T, 1, false, true); }  \
        else if (slices == 2) { if (nt) GO3(T, 2, true, false); else GO3(T, 2, false, false); } \
        else if (slices == 4) { if (nt) GO3(T, 4, true, false); else GO3(T, 4, false, false); } \
        else { if (nt) GO3(T, 1, true, false); else GO3(T, 1, false, false); }          \
    } while (0)
    if (threads == 256) GO3T(256);
    else if (threads == 384) GO3T(384);
    else if (threads == 512) GO3T(512);
    else GO3T(1024);
#undef GO3T
#undef GO3
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
