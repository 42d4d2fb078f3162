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
// Frees synchronise the device first (a caching allocator would recycle in stream order; this one unmaps).
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
        GA_CHECK(hipMemGetAllocationGranularity(&g_gran, &p, hipMemAllocationGranularityMinimum));
        const char* mode = std::getenv("FIERY_GUARD");
        g_before = mode && std::strcmp(mode, "before") == 0;
        std::fprintf(stderr, "guard_alloc: granularity %zu bytes, guard %s every tensor\n", g_gran, g_before ? "BEFORE" : "AFTER");
    }
    Block b;
    b.mapped = (static_cast<size_t>(size) + g_gran - 1) / g_gran * g_gran;
    b.reserved = b.mapped + 2 * g_gran;
    GA_CHECK(hipMemAddressReserve(&b.va, b.reserved, g_gran, nullptr, 0));
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
    GA_CHECK(hipMemAddressFree(b.va, b.reserved));
    g_live -= b.mapped;
}

// which tensor does a faulting address belong to?  (call from a debugger / ctypes after a fault report)
extern "C" void guard_report(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    std::fprintf(stderr, "guard_alloc: %zu allocations so far, %zu live, %.1f MB mapped now, %.1f MB at the peak\n", g_count,
                 g_blocks.size(), g_live / 1048576.0, g_peak / 1048576.0);
}
