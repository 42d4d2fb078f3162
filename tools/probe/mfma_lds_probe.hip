// Round 4 probe: what does feeding the fp32 MFMAs from LDS cost?  Every wavefront runs groups of eight
// v_mfma_f32_32x32x2_f32 (two accumulators alternating, as the conv kernel's 64 x 128 tile) whose operands come from R
// ds_read_b128 per group (R = 0: registers only; the conv loop has R = 3), optionally an s_barrier every 32 MFMAs (the conv
// loop's stage), optionally W ds_write_b128 per 32 MFMAs (its 6), at 1 / 2 / 3 workgroups per CU (LDS-limited like the kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_lds_probe.hip -o /tmp/mfma_lds_probe && /tmp/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));

template <int R, bool BARRIER, int W, int V = 0>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ rnd, float* __restrict__ out, int iters, int lds_floats) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < lds_floats; i += 256) lds[i] = rnd[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float4* l4 = reinterpret_cast<const float4*>(lds);
    float4* w4 = reinterpret_cast<float4*>(lds);
    v16f acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float4 a = l4[lane], b0 = l4[lane + 64], b1 = l4[lane + 128];
    float vv[8];
    for (int v = 0; v < 8; ++v) vv[v] = a.x + v;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {                 // four groups of eight MFMAs = one "stage" of 32
            float4 na = a, nb0 = b0, nb1 = b1;
            if (R >= 1) na = l4[lane + 64 * g];
            if (R >= 2) nb0 = l4[lane + 64 * g + 256];
            if (R >= 3) nb1 = l4[lane + 64 * g + 512];
            float4 extra = a;
            if (R >= 4) extra = l4[lane + 64 * g + 768];
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            if (W > 0 && g < W) w4[lane + 64 * g + 1024 + 256 * (threadIdx.x >> 6)] = extra;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(R >= 4 ? extra.x : a.w, b1.w, acc1, 0, 0, 0);
            if (V > 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) vv[v] = __builtin_fmaf(vv[v], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
            a = na;  b0 = nb0;  b1 = nb1;
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int v = 0; v < 8; ++v) s += vv[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, bool BARRIER, int W, int V = 0>
void run(int n_cu, const float* rnd, float* out, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int per_cu : {1, 2, 3}) {
        const int lds_bytes = per_cu == 1 ? 150000 : (per_cu == 2 ? 80000 : 49152);      // LDS-limited residency, like the conv kernel
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<R, BARRIER, W, V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        const int blocks = n_cu * per_cu, iters = 1500;
        auto launch = [&]() { hipLaunchKernelGGL((k_probe<R, BARRIER, W, V>), dim3(blocks), dim3(256), lds_bytes, 0, rnd, out, iters, 8192); };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 5.0 * blocks * 4 * iters * 32 * 4096.0;
        printf("%-58s %d wg/CU: %7.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)\n", what, per_cu, ms * 1e3 / 5, flops / (ms * 1e-3) / 1e12,
               flops / (ms * 1e-3) / 1e12 / 157.3 * 100);
    }
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<float> h(4096);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *rnd, *out;
    hipMalloc(&rnd, 4096 * 4);
    hipMalloc(&out, sizeof(float) * n_cu * 3 * 256);
    hipMemcpy(rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, false, 0>(n_cu, rnd, out, "MFMA only (operands in registers)");
    run<0, true, 0>(n_cu, rnd, out, "MFMA + barrier per 32");
    run<1, false, 0>(n_cu, rnd, out, "MFMA + 1 ds_read_b128 per 8");
    run<2, false, 0>(n_cu, rnd, out, "MFMA + 2 ds_read_b128 per 8");
    run<3, false, 0>(n_cu, rnd, out, "MFMA + 3 ds_read_b128 per 8 (the conv loop's reads)");
    run<4, false, 0>(n_cu, rnd, out, "MFMA + 4 ds_read_b128 per 8");
    run<3, true, 0>(n_cu, rnd, out, "MFMA + 3 reads per 8 + barrier per 32");
    run<4, true, 4>(n_cu, rnd, out, "MFMA + 4 reads per 8 + 4 ds_write_b128 + barrier per 32");
    run<3, true, 2>(n_cu, rnd, out, "MFMA + 3 reads per 8 + 2 ds_write_b128 + barrier per 32");
    run<0, false, 0, 2>(n_cu, rnd, out, "MFMA + 2 VALU per 8");
    run<0, false, 0, 8>(n_cu, rnd, out, "MFMA + 8 VALU per 8");
    run<3, true, 2, 2>(n_cu, rnd, out, "MFMA + 3 reads per 8 + 2 writes + barrier per 32 + 2 VALU per 8");
    return 0;
}
