// Tuning probe: does VALU work issued between fp32 MFMAs overlap with them on gfx950, or does it take matrix-pipe time?
// Each wave runs ITER x 16 v_mfma_f32_32x32x2_f32 (4 accumulators) with K independent VALU instructions after every MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_probe.hip -o mfma_valu_probe && ./mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
    v16f acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[16];
    int iv[16];
    for (int k = 0; k < 16; ++k) { v[k] = a + k; iv[k] = threadIdx.x + k; }
    __shared__ float lds[4096];
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);                       // fp32 VALU
                else if (KIND == 1) iv[k] = (iv[k] ^ (iv[k] >> 3)) + 7;                          // integer VALU (2 ops)
                else if (KIND == 2) v[k] += lds[(threadIdx.x * 4 + k * 64 + it) & 4095];          // LDS read + add
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int k = 0; k < 16; ++k) s += v[k] + iv[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int K, int KIND>
void run(const char* name, int blocks) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma = 16.0 * iters;
    const double flops = mfma * 4096.0 * 4 * blocks;     // 4 waves per block
    printf("%-14s K=%2d blocks=%4d: %8.1f cycles per MFMA (wave 0), %7.3f ms, %6.1f TFLOP/s\n", name, K, blocks, c / mfma, ms,
           flops / ms / 1e9);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0, 0>("none", blocks);
        run<2, 0>("fp32 valu", blocks);
        run<4, 0>("fp32 valu", blocks);
        run<8, 0>("fp32 valu", blocks);
        run<12, 0>("fp32 valu", blocks);
        run<16, 0>("fp32 valu", blocks);
        run<2, 1>("int valu x2", blocks);
        run<4, 1>("int valu x2", blocks);
        run<8, 1>("int valu x2", blocks);
        run<1, 2>("lds read+add", blocks);
        run<2, 2>("lds read+add", blocks);
        run<4, 2>("lds read+add", blocks);
    }
    return 0;
}
