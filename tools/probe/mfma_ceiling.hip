// Round 4 probe: what the fp32 matrix pipe of THIS box sustains - v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 back to back
// on every SIMD of every CU, random operands (power is data dependent), nothing else in the loop - and the clock it runs at
// while doing so (shader cycles from s_memtime against the 100 MHz wall clock).  The denominator question of DESIGN section 4:
// is 157.3 TFLOP/s (2.4 GHz) reachable under sustained matrix load, or does the chip settle lower?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ rnd, float* __restrict__ out, int iters,
                                              unsigned long long* __restrict__ clk) {
    float a[8], b[8];
    for (int k = 0; k < 8; ++k) {
        a[k] = rnd[(threadIdx.x * 8 + k) & 4095];
        b[k] = rnd[(threadIdx.x * 8 + k + 2048) & 4095];
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float s = 0.f;
    if (KIND == 0) {
        v16f acc[4];
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 7], b[(m + 3) & 7], acc[m & 3], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {
        v4f acc[8];
        for (int t = 0; t < 8; ++t)
            for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 7], b[(m + 3) & 7], acc[m & 7], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t)
            for (int r = 0; r < 4; ++r) s += acc[t][r];
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

int main(int argc, char** argv) {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<float> h(4096);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *rnd, *out;
    unsigned long long* clk;
    const int max_blocks = n_cu * 4;
    hipMalloc(&rnd, 4096 * 4);
    hipMalloc(&out, sizeof(float) * max_blocks * 256);
    hipMalloc(&clk, 16 * max_blocks);
    hipMemcpy(rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%d CUs; peak at 2.4 GHz = %.1f TFLOP/s (64 flop/clk/SIMD)\n", n_cu, n_cu * 4 * 64 * 2.4e9 / 1e12);
    for (int kind = 0; kind < 2; ++kind)
        for (int wps = 1; wps <= 3; ++wps)
            for (int iters : {600, 6000, 60000}) {
                const int blocks = n_cu * wps;                     // 256-thread blocks: one wavefront per SIMD each
                const double flops = (double)blocks * 4 * iters * (kind == 0 ? 16 * 4096.0 : 32 * 2048.0);
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL((k_mfma<0>), dim3(blocks), dim3(256), 0, 0, rnd, out, iters, clk);
                    else hipLaunchKernelGGL((k_mfma<1>), dim3(blocks), dim3(256), 0, 0, rnd, out, iters, clk);
                };
                const int reps = iters >= 60000 ? 3 : (iters >= 6000 ? 10 : 50);
                for (int r = 0; r < 2; ++r) launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < reps; ++r) launch();
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> c(2 * blocks);
                hipMemcpy(c.data(), clk, 16 * blocks, hipMemcpyDeviceToHost);
                double cyc = 0, wall = 0;
                for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
                const double ghz = cyc / (wall / 100e6) / 1e9;     // shader cycles per second of wall clock
                const double n_mfma = (double)iters * (kind == 0 ? 16 : 32);
                printf("%s  %d wave/SIMD  %6d iters: %8.1f us per launch  %6.1f TFLOP/s (%.1f%% of 157.3)  clock %.3f GHz  %.1f cycles per MFMA and wave\n",
                       kind == 0 ? "32x32x2 " : "16x16x4 ", wps, iters, ms * 1e3 / reps, flops * reps / (ms * 1e-3) / 1e12,
                       flops * reps / (ms * 1e-3) / 1e12 / 157.3 * 100, ghz, cyc / blocks / n_mfma);
            }
    return 0;
}
