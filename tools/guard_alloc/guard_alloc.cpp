// TEST TOOL (never part of the product path): a PyTorch pluggable device allocator that puts every tensor against an
// UNMAPPED page of GPU virtual address space, so that a kernel reading or writing past the end (FIERY_GUARD=after, the
// default) or before the start (FIERY_GUARD=before) of any tensor raises "Memory access fault by GPU" at once - the GPU
// counterpart of tests/sim/guard_check.py's mmap/mprotect harness, valid for vendor kernels (MIOpen, rocBLAS, ATen) too.
//
//   reserve  [ guard | data rounded up to the mapping granularity | guard ]   (hipMemAddressReserve)
//   map only the middle part                                                  (hipMemCreate + hipMemMap + hipMemSetAccess)
//   after:   the tensor ENDS at the upper guard (start 16-byte aligned: a tensor whose size is a multiple of 16 bytes has
//            no slack at all)          before:  the tensor STARTS at the lower guard
//
// Frees synchronise the device first (a caching allocator would recycle in stream order; this one unmaps).  The address
// RESERVATION of a freed tensor is kept: re-reserving a range that was just unmapped gave wrong data on ROCm 7.0.2 /
// MI355X (tools/guard_alloc/selftest.py fails with FIERY_GUARD_FREE_VA=1: stale translations) - and with addresses never
// reused a use-after-free faults as well.
// Build:  hipcc -O1 -fPIC -shared tools/guard_alloc/guard_alloc.cpp -o tools/guard_alloc/libguard_alloc.so
// Use:    tools/guard_alloc/run_guarded.py
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct Block {
    void* va;
    size_t reserved, mapped;
    hipMemGenericAllocationHandle_t handle;
};
std::mutex g_mutex;
std::unordered_map<void*, Block> g_blocks;
size_t g_gran = 0;
size_t g_live = 0, g_peak = 0, g_count = 0;
bool g_before = false;
// virtual addresses are never handed out twice (FIERY_GUARD_ARENA_GB of address space reserved once, carved
// front to back): a freed tensor's pages are unmapped and stay unmapped, so that use-after-free faults as well and no
// translation of an old mapping can ever be mistaken for a new one
char* g_arena = nullptr;
size_t g_arena_size = 0, g_arena_used = 0;
bool g_recommended = false, g_arena_on = true;

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}
#define GA_CHECK(call)                         \
    do {                                       \
        hipError_t e_ = (call);                \
        if (e_ != hipSuccess) die(#call, e_);  \
    } while (0)

hipMemAllocationProp props(int device) {
    hipMemAllocationProp p;
    std::memset(&p, 0, sizeof(p));
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    hipMemAllocationProp p = props(device);
    if (g_gran == 0) {
        const char* rec = std::getenv("FIERY_GUARD_GRANULARITY");
        g_recommended = rec && std::strcmp(rec, "recommended") == 0;
        GA_CHECK(hipMemGetAllocationGranularity(&g_gran, &p, g_recommended ? hipMemAllocationGranularityRecommended
                                                                             : hipMemAllocationGranularityMinimum));
        const char* mode = std::getenv("FIERY_GUARD");
        g_before = mode && std::strcmp(mode, "before") == 0;
        const char* arena = std::getenv("FIERY_GUARD_ARENA_GB");
        const size_t gb = arena ? std::strtoull(arena, nullptr, 10) : 0;
        g_arena_on = gb > 0;
        if (g_arena_on) {
            g_arena_size = gb << 30;
            void* va = nullptr;
            GA_CHECK(hipMemAddressReserve(&va, g_arena_size, g_gran, nullptr, 0));
            g_arena = static_cast<char*>(va);
        }
        std::fprintf(stderr, "guard_alloc: granularity %zu bytes, guard %s every tensor, %s\n", g_gran, g_before ? "BEFORE" : "AFTER",
                     g_arena_on ? "addresses never reused" : "addresses reserved per tensor");
    }
    Block b;
    b.mapped = (static_cast<size_t>(size) + g_gran - 1) / g_gran * g_gran;
    b.reserved = b.mapped + 2 * g_gran;
    if (g_arena_on) {
        if (g_arena_used + b.reserved > g_arena_size) die("address arena exhausted", hipErrorOutOfMemory);
        b.va = g_arena + g_arena_used;
        g_arena_used += b.reserved;
    } else {
        GA_CHECK(hipMemAddressReserve(&b.va, b.reserved, g_gran, nullptr, 0));
    }
    GA_CHECK(hipMemCreate(&b.handle, b.mapped, &p, 0));
    char* data = static_cast<char*>(b.va) + g_gran;
    GA_CHECK(hipMemMap(data, b.mapped, 0, b.handle, 0));
    hipMemAccessDesc access;
    std::memset(&access, 0, sizeof(access));
    access.location = p.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    GA_CHECK(hipMemSetAccess(data, b.mapped, &access, 1));
    const size_t size16 = (static_cast<size_t>(size) + 15) / 16 * 16;
    void* ptr = g_before ? data : data + (b.mapped - size16);
    g_blocks[ptr] = b;
    g_live += b.mapped;
    ++g_count;
    if (g_live > g_peak) g_peak = g_live;
    return ptr;
}

extern "C" void guard_free(void* ptr, ssize_t, int, hipStream_t) {
    if (!ptr) return;
    GA_CHECK(hipDeviceSynchronize());          // nothing may still be using the mapping
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_blocks.find(ptr);
    if (it == g_blocks.end()) {
        std::fprintf(stderr, "guard_alloc: free of an unknown pointer %p\n", ptr);
        std::abort();
    }
    Block b = it->second;
    g_blocks.erase(it);
    char* data = static_cast<char*>(b.va) + g_gran;
    GA_CHECK(hipMemUnmap(data, b.mapped));
    GA_CHECK(hipMemRelease(b.handle));
    // the reservation itself is kept (FIERY_GUARD_FREE_VA=1 returns it): the runtime then never hands the range out
    // again, so a freed tensor's addresses stay unmapped - use-after-free faults too, and no stale translation of an old
    // mapping can be mistaken for a new one
    static const bool free_va = std::getenv("FIERY_GUARD_FREE_VA") != nullptr;
    if (!g_arena_on && free_va) GA_CHECK(hipMemAddressFree(b.va, b.reserved));
    g_live -= b.mapped;
}

// which tensor does a faulting address belong to?  (call from a debugger / ctypes after a fault report)
extern "C" void guard_report(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    std::fprintf(stderr, "guard_alloc: %zu allocations so far, %zu live, %.1f MB mapped now, %.1f MB at the peak\n", g_count,
                 g_blocks.size(), g_live / 1048576.0, g_peak / 1048576.0);
}
