// TEST INFRASTRUCTURE: a tiny CPU stand-in for <hip/hip_runtime.h>.
//
// The product kernels under fiery_amd/csrc are written for gfx950 only and contain no host/device
// dual paths.  To exercise their index arithmetic, LDS tiling and MFMA fragment mapping in the
// GPU-less test tier, tests/sim/build_sim.py compiles those same .hip sources with g++ against THIS
// header (it shadows the real one through the include path).  One cooperative fiber plays one work-item;
// __syncthreads() parks fibers until the workgroup has arrived; the f32 MFMA builtin is emulated with the lane->element
// mapping documented for gfx950 (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row=(r&3)+8*(r>>2)+4*(l>>5),
// col=l&31).  It is slow, only meant for tiny problems, and never part of a shipped path.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
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
struct int2 { int x, y; };
inline int2 make_int2(int x, int y) { return {x, y}; }
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "sim"; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { __builtin_memset(p, v, n); return hipSuccess; }
// a 4-CU device: small test problems then span several rounds of workgroups, like the real sizes do on 256 CUs
inline hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = 4; return hipSuccess; }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// One work-item = one cooperative fiber (ucontext) on the calling OS thread; a workgroup's fibers are
// scheduled round-robin and only switch at barriers (__syncthreads, the wave-level exchange inside the
// emulated MFMA) or when they finish.  Workgroups run one after another.
namespace hipsim {
constexpr size_t kStack = 256 * 1024;

struct Barrier {
    int arrived = 0, generation = 0, alive = 0;
};
struct Tls {
    uint3_sim tid, bid;
    dim3 bdim, gdim;
    void* dyn_smem = nullptr;
    int lane = 0, wave = 0;
};
struct Fiber {
    ucontext_t ctx;
    Tls tls;
    bool done = false;
    const int* wait_gen = nullptr;   // parked until *wait_gen != wait_val
    int wait_val = 0;
};
struct Run {
    std::vector<Fiber> fibers;
    std::vector<Barrier> waves;
    std::vector<std::vector<float>> wave_a, wave_b;
    Barrier block;
    ucontext_t scheduler;
    int current = 0;
    std::function<void()> body;
};
inline Run*& run_ptr() {
    static Run* r = nullptr;
    return r;
}
inline Tls& tls() { return run_ptr()->fibers[run_ptr()->current].tls; }

inline void park(const int* gen, int val) {
    Run& r = *run_ptr();
    Fiber& f = r.fibers[r.current];
    f.wait_gen = gen;
    f.wait_val = val;
    swapcontext(&f.ctx, &r.scheduler);
}
inline void barrier_wait(Barrier& b) {
    const int gen = b.generation;
    if (++b.arrived == b.alive) {
        b.arrived = 0;
        ++b.generation;
    } else {
        park(&b.generation, gen);
    }
}
inline void barrier_leave(Barrier& b) {
    if (--b.alive > 0 && b.arrived == b.alive) {   // the others were only waiting for this one
        b.arrived = 0;
        ++b.generation;
    }
}
inline void trampoline() {
    Run& r = *run_ptr();
    r.body();
    Fiber& f = r.fibers[r.current];
    f.done = true;
    barrier_leave(r.waves[f.tls.wave]);
    barrier_leave(r.block);
    swapcontext(&f.ctx, &r.scheduler);
}
inline char* stack_pool(size_t n_fibers) {
    static char* pool = nullptr;
    static size_t cap = 0;
    if (n_fibers > cap) {
        free(pool);
        if (posix_memalign(reinterpret_cast<void**>(&pool), 4096, n_fibers * kStack)) abort();
        cap = n_fibers;
    }
    return pool;
}

template <typename Kernel, typename... Args>
void launch(Kernel kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    const unsigned n_threads = block.x * block.y * block.z;
    const unsigned n_waves = (n_threads + 63) / 64;
    void* smem = nullptr;
    if (posix_memalign(&smem, 64, shmem ? shmem : 64)) abort();
    char* stacks = stack_pool(n_threads);
    Run run;
    run.body = [&]() { kernel(args...); };
    Run* previous = run_ptr();
    run_ptr() = &run;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                run.fibers.assign(n_threads, Fiber());
                run.waves.assign(n_waves, Barrier());
                run.wave_a.assign(n_waves, std::vector<float>(64 * 8));     // (8 values per lane: the bf16 MFMA's operands)
                run.wave_b.assign(n_waves, std::vector<float>(64 * 8));
                run.block = Barrier();
                run.block.alive = static_cast<int>(n_threads);
                for (unsigned t = 0; t < n_threads; ++t) {
                    Fiber& f = run.fibers[t];
                    f.tls.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    f.tls.bid = {bx, by, bz};
                    f.tls.bdim = block;
                    f.tls.gdim = grid;
                    f.tls.dyn_smem = smem;
                    f.tls.lane = static_cast<int>(t % 64);
                    f.tls.wave = static_cast<int>(t / 64);
                    run.waves[t / 64].alive++;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = stacks + static_cast<size_t>(t) * kStack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, reinterpret_cast<void (*)()>(&trampoline), 0);
                }
                unsigned remaining = n_threads;
                while (remaining) {
                    for (unsigned t = 0; t < n_threads; ++t) {
                        Fiber& f = run.fibers[t];
                        if (f.done) continue;
                        if (f.wait_gen && *f.wait_gen == f.wait_val) continue;   // still parked
                        f.wait_gen = nullptr;
                        run.current = static_cast<int>(t);
                        swapcontext(&run.scheduler, &f.ctx);
                        if (f.done) --remaining;
                    }
                }
            }
    run_ptr() = previous;
    free(smem);
}
}  // namespace hipsim

#define threadIdx (::hipsim::tls().tid)
#define blockIdx (::hipsim::tls().bid)
#define blockDim (::hipsim::tls().bdim)
#define gridDim (::hipsim::tls().gdim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipsim::launch(kernel, grid, block, shmem, __VA_ARGS__)

inline void __syncthreads() { ::hipsim::barrier_wait(::hipsim::run_ptr()->block); }

using std::max;
using std::min;

inline float atomicAdd(float* addr, float v) {
    const float old = *addr;
    *addr = old + v;
    return old;
}
inline int atomicAdd(int* addr, int v) {
    const int old = *addr;
    *addr = old + v;
    return old;
}
inline unsigned long long atomicAdd(unsigned long long* addr, unsigned long long v) {
    const unsigned long long old = *addr;
    *addr = old + v;
    return old;
}

inline int atomicExch(int* addr, int v) {
    const int old = *addr;
    *addr = v;
    return old;
}

inline unsigned atomicOr(unsigned* addr, unsigned v) {
    const unsigned old = *addr;
    *addr = old | v;
    return old;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
// wave-wide vote: bit l of the result is lane l's predicate
inline unsigned long long __ballot(int predicate) {
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    float* flags = run.wave_a[t.wave].data();
    flags[t.lane] = predicate ? 1.f : 0.f;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    unsigned long long mask = 0;
    for (int l = 0; l < 64; ++l)
        if (flags[l] != 0.f) mask |= 1ull << l;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    return mask;
}

// wave-wide exchange: every lane of the wavefront calls it; lane l receives lane (l ^ mask)'s value
inline float __shfl_xor(float v, int mask) {
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    float* slot = run.wave_a[t.wave].data();
    slot[t.lane] = v;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    const float got = slot[(t.lane ^ mask) & 63];
    ::hipsim::barrier_wait(run.waves[t.wave]);
    return got;
}

inline int __shfl_xor(int v, int mask) {
    float f;
    __builtin_memcpy(&f, &v, 4);
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    int* slot = reinterpret_cast<int*>(run.wave_a[t.wave].data());
    slot[t.lane] = v;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    const int got = slot[(t.lane ^ mask) & 63];
    ::hipsim::barrier_wait(run.waves[t.wave]);
    return got;
}
// lane l receives lane (l - delta)'s value; lanes below `delta` keep their own (HIP's __shfl_up)
inline int __shfl_up(int v, unsigned delta) {
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    int* slot = reinterpret_cast<int*>(run.wave_a[t.wave].data());
    slot[t.lane] = v;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    const int got = t.lane >= static_cast<int>(delta) ? slot[t.lane - static_cast<int>(delta)] : v;
    ::hipsim::barrier_wait(run.waves[t.wave]);
    return got;
}
inline int __ffs(int v) { return __builtin_ffs(v); }

// f32 MFMA 32x32x2: D = A(32x2) . B(2x32) + C, one wave.
typedef float hipsim_v16f __attribute__((vector_size(64)));
#define HIP_SYMBOL(x) (&(x))
enum hipMemcpyKind { hipMemcpyHostToDevice = 1 };
inline hipError_t hipMemcpyToSymbolAsync(void* symbol, const void* src, size_t n, size_t offset, hipMemcpyKind, hipStream_t) {
    __builtin_memcpy(static_cast<char*>(symbol) + offset, src, n);
    return hipSuccess;
}
// raw buffer loads as gfx950 executes them (measured: tools/probe/buffer_oob_probe.hip, profiles/r3_buffer_oob_probe.txt):
// a descriptor is (base, size in bytes); the range check covers the vector offset PLUS the scalar offset (both taken as
// unsigned, the sum not wrapped) and is applied per dword - a 16-byte load that straddles the end returns its leading
// dwords and zeros; everything outside reads 0 and touches no memory.
struct __amdgpu_buffer_rsrc_t { const char* base; unsigned num; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num, int) {
    return {static_cast<const char*>(p), static_cast<unsigned>(num)};
}
template <int DWORDS>
inline void hipsim_buffer_load(__amdgpu_buffer_rsrc_t r, int voff, int soff, unsigned* dst) {
    const unsigned long long off = static_cast<unsigned long long>(static_cast<unsigned>(voff)) + static_cast<unsigned>(soff);
    for (int i = 0; i < DWORDS; ++i) {
        dst[i] = 0;
        if (off + 4ull * i + 4ull <= r.num) __builtin_memcpy(&dst[i], r.base + off + 4 * i, 4);
    }
}
typedef unsigned hipsim_v4u __attribute__((vector_size(16)));
inline hipsim_v4u __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    unsigned d[4];
    hipsim_buffer_load<4>(r, voff, soff, d);
    return hipsim_v4u{d[0], d[1], d[2], d[3]};
}
typedef unsigned hipsim_v2u __attribute__((vector_size(8)));
inline hipsim_v2u __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    unsigned d[2];
    hipsim_buffer_load<2>(r, voff, soff, d);
    return hipsim_v2u{d[0], d[1]};
}
inline void __builtin_amdgcn_raw_buffer_store_b128(hipsim_v4u v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    const unsigned long long off = static_cast<unsigned long long>(static_cast<unsigned>(voff)) + static_cast<unsigned>(soff);
    for (int i = 0; i < 4; ++i)
        if (off + 4ull * i + 4ull <= r.num) {
            const unsigned word = v[i];
            __builtin_memcpy(const_cast<char*>(r.base) + off + 4 * i, &word, 4);
        }
}
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    unsigned d[1];
    hipsim_buffer_load<1>(r, voff, soff, d);
    return d[0];
}
// correctly rounded single operations (no contraction): plain IEEE arithmetic on the host
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline unsigned long long clock64() { return 0; }
inline unsigned long long wall_clock64() { return 0; }
// instruction-scheduling hints have no effect on results
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      // only ever applied to wave-uniform values
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_exp2f(float x) { return __builtin_exp2f(x); }
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}

inline hipsim_v16f __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipsim_v16f c, int, int, int) {
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    const int l = t.lane, w = t.wave;
    float* wa = run.wave_a[w].data();
    float* wb = run.wave_b[w].data();
    wa[l] = a;
    wb[l] = b;
    ::hipsim::barrier_wait(run.waves[w]);
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = std::fmaf(wa[i], wb[j], acc);             // k = 0 : lanes 0..31
        acc = std::fmaf(wa[i + 32], wb[j + 32], acc);   // k = 1 : lanes 32..63
        c[r] = acc;
    }
    ::hipsim::barrier_wait(run.waves[w]);
    return c;
}
