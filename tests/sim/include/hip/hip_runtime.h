// TEST INFRASTRUCTURE: a tiny CPU stand-in for <hip/hip_runtime.h>.
//
// The product kernels under fiery_amd/csrc are written for gfx950 only and contain no host/device
// dual paths.  To exercise their index arithmetic, LDS tiling and MFMA fragment mapping in the
// GPU-less test tier, tests/sim/build_sim.py compiles those same .hip sources with g++ against THIS
// header (it shadows the real one through the include path).  One std::thread plays one work-item;
// __syncthreads() is a std::barrier; the f32 MFMA builtin is emulated with the lane->element
// mapping documented for gfx950 (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row=(r&3)+8*(r>>2)+4*(l>>5),
// col=l&31).  It is slow, only meant for tiny problems, and never part of a shipped path.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define ext_vector_type(n) vector_size(4 * (n))
#define HIP_DYNAMIC_SHARED(type, name) type* name = reinterpret_cast<type*>(::hipsim::tls().dyn_smem);

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_sim { unsigned x, y, z; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "sim"; }

namespace hipsim {
struct WaveShared {
    float a[64], b[64];
    std::barrier<> bar{64};
    explicit WaveShared(int lanes) : bar(lanes) {}
};
struct Tls {
    uint3_sim tid, bid;
    dim3 bdim, gdim;
    void* dyn_smem = nullptr;
    std::barrier<>* block_bar = nullptr;
    WaveShared* wave = nullptr;
    int lane = 0;
};
inline Tls& tls() {
    static thread_local Tls t;
    return t;
}

template <typename Kernel, typename... Args>
void launch(Kernel kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    const unsigned n_threads = block.x * block.y * block.z;
    const unsigned n_waves = (n_threads + 63) / 64;
    void* smem = nullptr;
    if (posix_memalign(&smem, 64, shmem ? shmem : 64)) abort();
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                std::barrier<> block_bar(n_threads);
                std::vector<std::unique_ptr<WaveShared>> waves;
                for (unsigned w = 0; w < n_waves; ++w) {
                    unsigned lanes = std::min(64u, n_threads - w * 64);
                    waves.emplace_back(new WaveShared(static_cast<int>(lanes)));
                }
                std::vector<std::thread> pool;
                pool.reserve(n_threads);
                for (unsigned t = 0; t < n_threads; ++t) {
                    pool.emplace_back([&, t, bx, by, bz]() {
                        Tls& c = tls();
                        c.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                        c.bid = {bx, by, bz};
                        c.bdim = block;
                        c.gdim = grid;
                        c.dyn_smem = smem;
                        c.block_bar = &block_bar;
                        c.wave = waves[t / 64].get();
                        c.lane = static_cast<int>(t % 64);
                        kernel(args...);
                        c.wave->bar.arrive_and_drop();
                        block_bar.arrive_and_drop();
                    });
                }
                for (auto& th : pool) th.join();
            }
    free(smem);
}
}  // namespace hipsim

#define threadIdx (::hipsim::tls().tid)
#define blockIdx (::hipsim::tls().bid)
#define blockDim (::hipsim::tls().bdim)
#define gridDim (::hipsim::tls().gdim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipsim::launch(kernel, grid, block, shmem, __VA_ARGS__)

inline void __syncthreads() { ::hipsim::tls().block_bar->arrive_and_wait(); }

using std::max;
using std::min;

inline float atomicAdd(float* addr, float v) {
    uint32_t* p = reinterpret_cast<uint32_t*>(addr);
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        std::memcpy(&f, &old, 4);
        float nf = f + v;
        uint32_t desired;
        std::memcpy(&desired, &nf, 4);
        if (__atomic_compare_exchange_n(p, &old, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
inline int atomicAdd(int* addr, int v) { return __atomic_fetch_add(addr, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* addr, unsigned long long v) {
    return __atomic_fetch_add(addr, v, __ATOMIC_RELAXED);
}

// f32 MFMA 32x32x2: D = A(32x2) . B(2x32) + C, one wave.
typedef float hipsim_v16f __attribute__((vector_size(64)));
inline hipsim_v16f __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipsim_v16f c, int, int, int) {
    ::hipsim::Tls& t = ::hipsim::tls();
    ::hipsim::WaveShared& w = *t.wave;
    const int l = t.lane;
    w.a[l] = a;
    w.b[l] = b;
    w.bar.arrive_and_wait();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = std::fmaf(w.a[i], w.b[j], acc);             // k = 0 : lanes 0..31
        acc = std::fmaf(w.a[i + 32], w.b[j + 32], acc);   // k = 1 : lanes 32..63
        c[r] = acc;
    }
    w.bar.arrive_and_wait();
    return c;
}
