// CEILING PROBE for the voxel-pooling op (`projection_to_birds_eye_view`, fiery/models/fiery.py:221-273), round 5.
//
// What does THIS box's memory system give a kernel that moves exactly the op's algorithmic bytes and does nothing else?
//   reads  = 4 C N_kept (the lifted values of in-grid points) + 12 N (geometry)         1,101.1 MB at baseline.yml batch 3
//   writes = 4 C X Y per frame (the BEV planes)                                            92.2 MB
// Two families, each swept over its launch parameters; the best time of all is the op's ceiling on this box:
//   (a) pure streaming: every workgroup reads its share of ONE contiguous buffer of the read bytes (non-temporal or plain,
//       grid-stride or slabs, 4 / 8 loads of 16 bytes in flight per lane) and writes its share of the output buffer;
//   (b) op-shaped: 576 (channel, frame) units, each streaming its channel's rows of the (n, C, D, H, W) lifted tensor in the
//       best pattern round 2 found (rows of a slice dealt to four lane groups, non-temporal; tools/probe/hbm_probe.hip
//       k_read_planes3) and then writing its 160,000-byte plane - the access shape the product kernel is bound to by the
//       encoder's native layout (reads ALL rows, kept or not: 1,114.8 + 52.3 MB).
// Nothing is computed: no ranks, no run sums, no LDS atomics, no descriptors, no prepass.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pool_ceiling.hip -o tools/probe/_bin/pool_ceiling && tools/probe/_bin/pool_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld16(const float4* p) {
    if (NT) {
        const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ void st16(float4* p, float4 v, bool nt) {
    if (nt) __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4*>(p));
    else *p = v;
}

// (a) streaming: reads n_r 16-byte words, writes n_w
template <int U, bool NT, bool SLABS>
__global__ void k_stream(const float4* __restrict__ x, long long n_r, float4* __restrict__ y, long long n_w, int nt_store, float* __restrict__ sink) {
    const long long nthreads = static_cast<long long>(gridDim.x) * blockDim.x;
    float acc = 0.f;
    if (SLABS) {
        const long long per = (n_r + gridDim.x - 1) / gridDim.x;
        const long long lo = per * blockIdx.x, hi = std::min(lo + per, n_r);
        long long i = lo + threadIdx.x;
        for (; i + (U - 1) * static_cast<long long>(blockDim.x) < hi; i += U * static_cast<long long>(blockDim.x)) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16<NT>(x + i + u * static_cast<long long>(blockDim.x));
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
        for (; i < hi; i += blockDim.x) acc += x[i].x;
    } else {
        long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
        for (; i + (U - 1) * nthreads < n_r; i += U * nthreads) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16<NT>(x + i + u * nthreads);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
        for (; i < n_r; i += nthreads) acc += x[i].x;
    }
    const float4 z = make_float4(acc, 0.f, 0.f, 0.f);
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_w; i += nthreads) st16(y + i, z, nt_store != 0);
    if (acc == 1.2345e-30f) sink[0] = acc;
}

// (b) op-shaped: unit = (channel c, frame f); lifted tensor (frames * 6, 64, 48, 28, 60) floats; geometry read as 12 N bytes
// spread over the units; then the unit's plane (40,000 floats) written
template <int T, bool NT>
__global__ __launch_bounds__(T) void k_units(const float4* __restrict__ x, const float4* __restrict__ geo, long long geo_per_unit4,
                                             float4* __restrict__ y, float* __restrict__ sink) {
    const int c = blockIdx.x % 64, f = blockIdx.x / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int n_waves = T / 64;
    const int g = lane >> 4, col = lane & 15;
    float acc = 0.f;
    if (col < 15)
        for (int s = wave; s < 6 * 48; s += n_waves) {
            const int cam = s / 48, d = s - cam * 48;
            const float4* p = x + ((((long long)f * 6 + cam) * 64 + c) * 48 + d) * 420 + g * 15 + col;
            float4 v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = ld16<NT>(p + j * 60);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
    const float4* gp = geo + static_cast<long long>(blockIdx.x) * geo_per_unit4;
    for (long long i = threadIdx.x; i < geo_per_unit4; i += T) acc += ld16<NT>(gp + i).x;
    float4* plane = y + static_cast<long long>(blockIdx.x) * 10000;
    const float4 z = make_float4(acc, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < 10000; i += T) plane[i] = z;
    if (acc == 1.2345e-30f) sink[0] = acc;
}

template <typename F>
float time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 9;                                   // baseline.yml batch 3: 9 frames
    const double kept = argc > 2 ? atof(argv[2]) : 0.9409;                             // N_kept / N of the synthetic rig (bench.py: kept_fraction)
    const long long N = 6ll * 48 * 28 * 60 * frames;                                   // points
    const long long lifted_all = 4ll * 64 * N, geo = 12ll * N, out = 4ll * 64 * 200 * 200 * frames;
    const long long r_bytes = static_cast<long long>(4.0 * 64 * N * kept) / 16 * 16 + geo, w_bytes = out;
    std::printf("pooling ceiling probe: %d frames, N = %lld points, kept %.4f: algorithmic reads %.1f MB + writes %.1f MB = %.1f MB\n", frames, N,
                kept, r_bytes / 1e6, w_bytes / 1e6, (r_bytes + w_bytes) / 1e6);
    float4 *x, *g, *y;
    float* sink;
    hipMalloc(&x, lifted_all);
    hipMalloc(&g, geo + (1 << 20));
    hipMalloc(&y, out);
    hipMalloc(&sink, 64);
    hipMemset(x, 0, lifted_all);
    hipMemset(g, 0, geo + (1 << 20));
    // throw-away: the clock settles
    time_us([&] { hipLaunchKernelGGL((k_stream<8, true, false>), dim3(2048), dim3(256), 0, 0, x, r_bytes / 16, y, w_bytes / 16, 0, sink); }, 50);
    float best = 1e9f;
    const char* best_what = "";
    static char names[256][160];
    int n_names = 0;
    auto report = [&](const char* what, float us, long long bytes) {
        std::printf("  %-100s %7.1f us  %7.1f GB/s\n", what, us, bytes / us / 1e3);
        if (us < best) {
            best = us;
            std::snprintf(names[n_names], sizeof names[0], "%s", what);
            best_what = names[n_names++];
        }
    };
    std::printf("(a) streaming the algorithmic bytes (%.1f MB read from one buffer, %.1f MB written):\n", r_bytes / 1e6, w_bytes / 1e6);
    for (int wgs : {1024, 2048, 4096, 8192})
        for (int threads : {256, 512})
            for (int nt = 0; nt < 2; ++nt)
                for (int slabs = 0; slabs < 2; ++slabs)
                    for (int u : {4, 8})
                        for (int nts = 0; nts < 2; ++nts) {
                            char what[160];
                            std::snprintf(what, sizeof what, "%5d workgroups x %3d threads, %s loads, %s, %d x 16 B in flight, %s stores", wgs, threads,
                                          nt ? "non-temporal" : "plain", slabs ? "slabs" : "grid-stride", u, nts ? "non-temporal" : "plain");
                            auto go = [&] {
#define GO(U, NT, SL) hipLaunchKernelGGL((k_stream<U, NT, SL>), dim3(wgs), dim3(threads), 0, 0, x, r_bytes / 16, y, w_bytes / 16, nts, sink)
                                if (u == 4) { if (nt) { if (slabs) GO(4, true, true); else GO(4, true, false); } else { if (slabs) GO(4, false, true); else GO(4, false, false); } }
                                else { if (nt) { if (slabs) GO(8, true, true); else GO(8, true, false); } else { if (slabs) GO(8, false, true); else GO(8, false, false); } }
#undef GO
                            };
                            const float us = time_us(go);
                            if (us < best * 1.03f) report(what, us, r_bytes + w_bytes);
                            else best = std::min(best, us);
                        }
    const float best_stream = best;
    std::printf("    best streaming pattern: %.1f us = %.1f GB/s (%s)\n", best_stream, (r_bytes + w_bytes) / best_stream / 1e3, best_what);
    std::printf("(b) op-shaped units (576 x frames/9 (channel, frame) units over the (n, C, D, H, W) tensor: ALL rows read, %.1f MB + geometry %.1f MB, planes written):\n",
                lifted_all / 1e6, geo / 1e6);
    float best_units = 1e9f;
    const int units = 64 * frames;
    const long long geo_per_unit4 = geo / 16 / units;
    for (int nt = 0; nt < 2; ++nt) {
#define GOU(T, LDS)                                                                                                                           \
    do {                                                                                                                                      \
        char what[160];                                                                                                                       \
        std::snprintf(what, sizeof what, "%4d threads, %d per CU (dynamic LDS %d B), %s loads", T, 163840 / LDS, LDS, nt ? "non-temporal" : "plain"); \
        auto go = [&] {                                                                                                                       \
            if (nt) hipLaunchKernelGGL((k_units<T, true>), dim3(units), dim3(T), LDS, 0, x, g, geo_per_unit4, y, sink);                        \
            else hipLaunchKernelGGL((k_units<T, false>), dim3(units), dim3(T), LDS, 0, x, g, geo_per_unit4, y, sink);                          \
        };                                                                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_units<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_units<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);         \
        const float us = time_us(go);                                                                                                         \
        best_units = std::min(best_units, us);                                                                                                \
        std::printf("  %-100s %7.1f us  %7.1f GB/s physical, %7.1f GB/s algorithmic\n", what, us, (lifted_all + geo + out) / us / 1e3,        \
                    (r_bytes + w_bytes) / us / 1e3);                                                                                          \
    } while (0)
        GOU(1024, 160000);
        GOU(512, 80000);
        GOU(384, 80000);
        GOU(256, 54000);
#undef GOU
    }
    const float ceiling = std::min(best_stream, best_units);
    std::printf("CEILING: %.1f us for the op's %.1f MB on this box = %.1f GB/s = %.3f of the 8 TB/s spec (streaming %.1f us, op-shaped %.1f us)\n", ceiling,
                (r_bytes + w_bytes) / 1e6, (r_bytes + w_bytes) / ceiling / 1e3, (r_bytes + w_bytes) / ceiling / 1e3 / 8000.0, best_stream, best_units);
    return 0;
}
