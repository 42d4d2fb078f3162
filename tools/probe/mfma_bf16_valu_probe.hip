// Probe (standalone): do vector-ALU instructions run in the shadow of bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8 passes) of the
// same wavefront / of the SIMD's other wavefront?  (For the fp32 MFMA they do not: tools/probe/mfma_valu_probe.hip.)
// Per loop iteration: two independent MFMAs and N vector instructions of the kind a bf16 split needs (v_and_b32,
// v_pk_add_f32, v_perm_b32) on registers the MFMAs do not touch.  W wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N, int FP32>
__global__ __launch_bounds__(256) void k(float* sink, int reps) {
    v16f acc0 = {0}, acc1 = {0};
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = static_cast<__bf16>(1.f + threadIdx.x * 1e-3f + j);  y[j] = static_cast<__bf16>(0.5f + j); }
    unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x12345u, c = a + 77u;
    v2f f = {1.f + threadIdx.x, 2.f}, g = {0.5f, 0.25f};
    for (int r = 0; r < reps; ++r) {
        if (FP32) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, g.x, acc0, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i % 3 == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a) : "v"(b));
            else if (i % 3 == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
            else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b) : "v"(a), "v"(c), "v"(c));
        }
    }
    float s = f.x + f.y + __uint_as_float(a) + __uint_as_float(b);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 1.2345f) sink[0] = s;
}

template <int N, int FP32>
void run(float* sink, int waves) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));  CHECK(hipEventCreate(&e1));
    const int reps = 20000, blocks = 256 * waves;
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<N, FP32>), dim3(blocks), dim3(256), 0, 0, sink, reps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    // cycles per iteration and SIMD at 2.4 GHz
    printf("%s  waves/SIMD %d  VALU per iteration %2d : %.3f ms  = %.1f ns per iteration and wavefront slot\n", FP32 ? "fp32 mfma x1" : "bf16 mfma x2", waves, N, best,
           best * 1e6 / reps);
}

int main() {
    float* sink;
    CHECK(hipMalloc(&sink, 64));
    for (int waves = 1; waves <= 2; ++waves) {
        run<0, 0>(sink, waves);  run<3, 0>(sink, waves);  run<6, 0>(sink, waves);  run<9, 0>(sink, waves);  run<12, 0>(sink, waves);  run<18, 0>(sink, waves);  run<24, 0>(sink, waves);
        run<0, 1>(sink, waves);  run<6, 1>(sink, waves);  run<12, 1>(sink, waves);
    }
    return 0;
}
