// Does a launch with a large by-value argument block cost a __amd_rocclr_copyBuffer dispatch?  (one appears per convolution
// launch in every rocprofv3 trace of the step: 75 per step, 4.75 us each)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/kernarg_probe.hip -o /tmp/kernarg_probe
//   rocprofv3 --kernel-trace --stats -d /tmp/kp -- /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
struct Args {
    float* out;
    int v[N];
};
template <int N>
__global__ void k_args(Args<N> a) {
    if (threadIdx.x == 0) a.out[blockIdx.x] = static_cast<float>(a.v[N - 1]);
}

template <int N>
void run(float* d, hipStream_t s, const char* how) {
    Args<N> a;
    a.out = d;
    for (int i = 0; i < N; ++i) a.v[i] = i;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_args<N>, dim3(4), dim3(64), 0, s, a);
    (void)hipStreamSynchronize(s);
    std::printf("%s: 200 launches with a %zu-byte argument block\n", how, sizeof(a));
}

int main() {
    float* d;
    (void)hipMalloc(&d, 1024);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    run<4>(d, s, "eager");
    run<60>(d, s, "eager");
    run<180>(d, s, "eager");
    run<250>(d, s, "eager");
    // the same inside a captured graph, replayed
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    {
        Args<180> a;
        a.out = d;
        for (int i = 0; i < 180; ++i) a.v[i] = i;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_args<180>, dim3(4), dim3(64), 0, s, a);
    }
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    std::printf("graph: 10 replays of 20 launches with a 728-byte argument block\n");
    return 0;
}
