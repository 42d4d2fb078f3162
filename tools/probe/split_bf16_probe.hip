// Probe (standalone; hipcc --offload-arch=gfx950 -O2 -o split_bf16_probe split_bf16_probe.hip): how close does a GEMM get
// to the fp64 result when fp32 operands are split into 1 / 2 / 3 bf16 terms and multiplied on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) - against the fp32 matrix instruction (v_mfma_f32_32x32x2_f32) on the
// same data - and what the two instructions cost per 32 x 32 x 16 block.
//   x1: a1.b1                                   (the bf16 operand mode)
//   x3: a1.b1 + a1.b2 + a2.b1                   (two terms each)
//   x6: x3 + a1.b3 + a2.b2 + a3.b1              (three terms each, products below 2^-24 |a||b| dropped)
//   x9: all nine
// "two accumulators": the correction products go into an accumulator of their own that is added at the end.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = static_cast<__bf16>(v);
    const float r = v - static_cast<float>(a);
    b = static_cast<__bf16>(r);
    const float r2 = r - static_cast<float>(b);
    c = static_cast<__bf16>(r2);
}

// A: [tiles][32][K] row-major, B: [tiles][K][32]; C: [tiles][mode][32][32]
__global__ void k_probe(const float* A, const float* B, float* C, int K, int modes) {
    const int t = blockIdx.x, l = threadIdx.x, m = l & 31, hi = l >> 5;
    const float* a = A + static_cast<size_t>(t) * 32 * K;
    const float* b = B + static_cast<size_t>(t) * K * 32;
    v16f acc32 = {0}, acc1 = {0}, acc3 = {0}, acc6 = {0}, acc9 = {0}, big = {0}, small6 = {0}, small9 = {0};
    for (int k = 0; k < K; k += 2) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m * K + k + hi], b[(k + hi) * 32 + m], acc32, 0, 0, 0);
    for (int k = 0; k < K; k += 16) {
        bf16x8 a1, a2, a3, b1, b2, b3;
        for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(a[m * K + k + 8 * hi + j], x, y, z);  a1[j] = x;  a2[j] = y;  a3[j] = z;
            split3(b[(k + 8 * hi + j) * 32 + m], x, y, z);  b1[j] = x;  b2[j] = y;  b3[j] = z;
        }
#define MM(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0)
        MM(a1, b1, acc1);
        MM(a1, b2, acc3);  MM(a2, b1, acc3);  MM(a1, b1, acc3);
        MM(a1, b3, acc6);  MM(a3, b1, acc6);  MM(a2, b2, acc6);  MM(a1, b2, acc6);  MM(a2, b1, acc6);  MM(a1, b1, acc6);
        MM(a3, b3, acc9);  MM(a2, b3, acc9);  MM(a3, b2, acc9);
        MM(a1, b3, acc9);  MM(a3, b1, acc9);  MM(a2, b2, acc9);  MM(a1, b2, acc9);  MM(a2, b1, acc9);  MM(a1, b1, acc9);
        MM(a1, b1, big);
        MM(a1, b3, small6);  MM(a3, b1, small6);  MM(a2, b2, small6);  MM(a1, b2, small6);  MM(a2, b1, small6);
        MM(a3, b3, small9);  MM(a2, b3, small9);  MM(a3, b2, small9);
        MM(a1, b3, small9);  MM(a3, b1, small9);  MM(a2, b2, small9);  MM(a1, b2, small9);  MM(a2, b1, small9);
    }
    const v16f two6 = big + small6, two9 = big + small9;
    const v16f* out[7] = {&acc32, &acc1, &acc3, &acc6, &acc9, &two6, &two9};
    for (int md = 0; md < modes; ++md)
        for (int i = 0; i < 16; ++i) {
            const int row = (i / 4) * 8 + hi * 4 + (i % 4);
            C[((static_cast<size_t>(t) * modes + md) * 32 + row) * 32 + m] = (*out[md])[i];
        }
}

// instruction cost: `reps` back-to-back dependent-free MFMAs per wavefront (four accumulators), four wavefronts per SIMD
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* sink, int reps) {
    v16f acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = static_cast<__bf16>(1.f + threadIdx.x * 1e-3f + j);  y[j] = static_cast<__bf16>(0.5f + j); }
    const float fa = 1.f + threadIdx.x * 1e-3f, fb = 0.5f;
    for (int r = 0; r < reps; ++r)
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[u], 0, 0, 0);
            else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[u], 0, 0, 0);
        }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) s += acc[u][i];
    if (s == 1.2345f) sink[0] = s;
}

int main() {
    const int tiles = 64, modes = 7;
    const char* names[modes] = {"fp32 mfma", "bf16 x1", "bf16 x3", "bf16 x6", "bf16 x9", "x6 two acc", "x9 two acc"};
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int relu = 0; relu < 2; ++relu)
        for (int K : {64, 576, 1152, 4608}) {
            std::vector<float> A(static_cast<size_t>(tiles) * 32 * K), B(static_cast<size_t>(tiles) * K * 32), C(static_cast<size_t>(tiles) * modes * 1024);
            for (auto& v : A) { v = nd(rng);  if (relu && v < 0) v = 0; }
            for (auto& v : B) v = nd(rng) / std::sqrt(static_cast<float>(K));
            float *dA, *dB, *dC;
            CHECK(hipMalloc(&dA, A.size() * 4));  CHECK(hipMalloc(&dB, B.size() * 4));  CHECK(hipMalloc(&dC, C.size() * 4));
            CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_probe, dim3(tiles), dim3(64), 0, 0, dA, dB, dC, K, modes);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double err2[modes] = {0}, errmax[modes] = {0}, ref2 = 0, f32_2 = 0, f32max = 0;
            for (int t = 0; t < tiles; ++t)
                for (int r = 0; r < 32; ++r)
                    for (int c = 0; c < 32; ++c) {
                        double s = 0;
                        float s32 = 0.f;
                        for (int k = 0; k < K; ++k) {
                            const float x = A[(static_cast<size_t>(t) * 32 + r) * K + k], y = B[(static_cast<size_t>(t) * K + k) * 32 + c];
                            s += static_cast<double>(x) * y;
                            s32 = std::fmaf(x, y, s32);
                        }
                        ref2 += s * s;
                        f32_2 += (s32 - s) * (s32 - s);
                        f32max = std::fmax(f32max, std::fabs(s32 - s));
                        for (int md = 0; md < modes; ++md) {
                            const double e = C[((static_cast<size_t>(t) * modes + md) * 32 + r) * 32 + c] - s;
                            err2[md] += e * e;
                            errmax[md] = std::fmax(errmax[md], std::fabs(e));
                        }
                    }
            const double n = static_cast<double>(tiles) * 1024, rms = std::sqrt(ref2 / n);
            printf("K = %4d  %s  rms(C) = %.3f   [rms error / rms(C), max error / rms(C)]\n", K, relu ? "A >= 0 (after ReLU)" : "A ~ N(0,1)", rms);
            printf("   %-12s %.3e  %.3e\n", "fp32 fma cpu", std::sqrt(f32_2 / n) / rms, f32max / rms);
            for (int md = 0; md < modes; ++md) printf("   %-12s %.3e  %.3e\n", names[md], std::sqrt(err2[md] / n) / rms, errmax[md] / rms);
            CHECK(hipFree(dA));  CHECK(hipFree(dB));  CHECK(hipFree(dC));
        }
    // instruction rates
    float* sink;
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));  CHECK(hipEventCreate(&e1));
    const int reps = 20000, blocks = 256 * 4;                  // four workgroups of four wavefronts per CU: four wavefronts per SIMD
    for (int mode = 0; mode < 2; ++mode)
        for (int it = 0; it < 3; ++it) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, sink, reps);
            else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, sink, reps);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = static_cast<double>(blocks) * 4 * reps * 4 * (mode == 0 ? 4096.0 : 32768.0);
            printf("%s: %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "v_mfma_f32_32x32x2_f32   " : "v_mfma_f32_32x32x16_bf16 ", ms, flops / ms * 1e-9);
        }
    return 0;
}
