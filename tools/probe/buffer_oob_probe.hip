// What does the gfx950 range check of a raw buffer load cover?  (the kernels rely on "outside the descriptor reads 0")
//   hipcc --offload-arch=gfx950 -O2 tools/probe/buffer_oob_probe.hip -o /tmp/buffer_oob_probe && /tmp/buffer_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(const float* data, int num_records, const int* voffs, const int* soffs, int n, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(data), 0, num_records, 0x00020000);
    for (int i = 0; i < n; ++i) {
        const int so = __builtin_amdgcn_readfirstlane(soffs[i]);
        auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voffs[i], so, 0);
        float f[4];
        __builtin_memcpy(f, &v, 16);
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = f[j];
    }
}

int main() {
    const int N = 4096;
    float* d;
    hipMalloc(&d, N * 4);
    float h[N];
    for (int i = 0; i < N; ++i) h[i] = 1000.f + i;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int voffs[] = {0, 240, 244, 256, 0, 128, 0, static_cast<int>(0x80000000u), 0, 1024, 16};
    const int soffs[] = {0, 0, 0, 0, 256, 128, 1024, 0, 240, -1024 + 0, 0x7ffffff0};
    const int n = sizeof(voffs) / 4;
    int *dv, *ds;
    float* dout;
    hipMalloc(&dv, sizeof(voffs)); hipMalloc(&ds, sizeof(soffs)); hipMalloc(&dout, n * 16);
    hipMemcpy(dv, voffs, sizeof(voffs), hipMemcpyHostToDevice);
    hipMemcpy(ds, soffs, sizeof(soffs), hipMemcpyHostToDevice);
    // descriptor: base = data + 256 floats (so that negative scalar offsets stay inside the allocation), 256 BYTES long
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d + 256, 256, dv, ds, n, dout);
    float out[64];
    hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost);
    std::printf("descriptor: 256 bytes = elements 1256 .. 1319\n");
    for (int i = 0; i < n; ++i)
        std::printf("voffset %11d  soffset %11d  -> %7.0f %7.0f %7.0f %7.0f\n", voffs[i], soffs[i], out[i * 4], out[i * 4 + 1], out[i * 4 + 2],
                    out[i * 4 + 3]);
    return 0;
}
