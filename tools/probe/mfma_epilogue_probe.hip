// Round 4 probe: what does a tile's EPILOGUE cost the fp32 matrix pipe when three workgroups share a CU?
// Every wavefront runs `tiles` tiles; a tile = 36 stages of 32 v_mfma_f32_32x32x2_f32 fed by 12 ds_read_b128, one barrier
// per stage (the 64 x 128 conv tile's K loop at K = 1152), then an epilogue of E vector instructions + 8 global stores:
//   kind 0: the epilogue after the K loop (the kernel as it is);
//   kind 1: the same E instructions dealt out between the MFMAs of the NEXT tile's K loop (a deferred epilogue).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_epilogue_probe.hip -o /tmp/mfma_epilogue_probe && /tmp/mfma_epilogue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef TILES
#define TILES 3
#endif

typedef float v16f __attribute__((ext_vector_type(16)));

template <int E, int KIND>
__global__ __launch_bounds__(256, 3) void k_tiles(const float* __restrict__ rnd, float* __restrict__ out, int tiles) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = rnd[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float4* l4 = reinterpret_cast<const float4*>(lds);
    v16f acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float4 a = l4[lane], b0 = l4[lane + 64], b1 = l4[lane + 128];
    float vv[8];
    for (int v = 0; v < 8; ++v) vv[v] = a.x + v;
    constexpr int PER_GROUP = KIND == 1 ? (E + 143) / 144 : 0;          // 144 groups of eight MFMAs per tile
    float* dst = out + (blockIdx.x * 256 + threadIdx.x) * 4;
    for (int t = 0; t < tiles; ++t) {
        for (int st = 0; st < 36; ++st) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 na = l4[lane + 64 * g], nb0 = l4[lane + 64 * g + 256], nb1 = l4[lane + 64 * g + 512];
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                if (KIND == 1) {
#pragma unroll
                    for (int v = 0; v < PER_GROUP; ++v) vv[v & 7] = __builtin_fmaf(vv[v & 7], 1.0001f, 0.5f);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a = na;  b0 = nb0;  b1 = nb1;
            }
            __syncthreads();
        }
        if (KIND == 0) {
#pragma unroll 8
            for (int v = 0; v < E; ++v) vv[v & 7] = __builtin_fmaf(vv[v & 7], 1.0001f, 0.5f);
        }
        if (E > 0) {
#pragma unroll
            for (int v = 0; v < 8; ++v) __builtin_nontemporal_store(vv[v] + acc0[v], dst + (v & 3) + 1024 * (v >> 2) * (t & 1));
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int v = 0; v < 8; ++v) s += vv[v];
    dst[0] = s;
}

template <int E, int KIND>
void run(int n_cu, const float* rnd, float* out, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int lds_bytes = 49152, blocks = n_cu * 3, tiles = TILES;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tiles<E, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    auto launch = [&]() { hipLaunchKernelGGL((k_tiles<E, KIND>), dim3(blocks), dim3(256), lds_bytes, 0, rnd, out, tiles); };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * tiles * 36 * 32 * 4096.0;
    printf("%-64s %7.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)\n", what, ms * 1e3 / 5, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3 * 100);
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<float> h(4096);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *rnd, *out;
    hipMalloc(&rnd, 4096 * 4);
    hipMalloc(&out, sizeof(float) * (n_cu * 3 * 256 * 4 + 4096));
    hipMemcpy(rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("three workgroups per CU, %d tiles each", TILES); printf("  (36 stages x 32 MFMAs + 12 ds_read_b128 + barrier), then per tile:\n");
    run<2000, 0>(n_cu, rnd, out, "(warm-up of the clocks: not a measurement)");
    run<0, 0>(n_cu, rnd, out, "no epilogue");
    run<250, 0>(n_cu, rnd, out, "epilogue of 250 vector instructions after the K loop");
    run<500, 0>(n_cu, rnd, out, "epilogue of 500");
    run<1000, 0>(n_cu, rnd, out, "epilogue of 1000");
    run<2000, 0>(n_cu, rnd, out, "epilogue of 2000");
    run<500, 1>(n_cu, rnd, out, "500 dealt out between the next tile's MFMAs");
    run<1000, 1>(n_cu, rnd, out, "1000 dealt out between the next tile's MFMAs");
    run<0, 0>(n_cu, rnd, out, "no epilogue (again)");
    run<1000, 0>(n_cu, rnd, out, "epilogue of 1000 (again)");
    return 0;
}
