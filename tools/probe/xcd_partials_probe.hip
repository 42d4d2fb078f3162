// Probe for the K-split of a convolution launch (NOTES.md, "K-split of a launch's last round"): partial output tiles handed
// from one workgroup to another through a workspace in HBM, where the two may sit on different XCDs (L2s not coherent).
// Not run yet - written at the end of round 4 when the GPU minutes were spent; the first thing to run in round 5:
//   hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_partials_probe.hip -o /tmp/xcd_partials_probe && /tmp/xcd_partials_probe
//
// PARTS workgroups per tile; each writes its 64 KB partial (128 x 128 fp32, full-row 16-byte stores as the conv epilogue would),
// takes a ticket on counter[tile]; the last arrival reads all PARTS partials back in part order, sums them and writes the
// tile out.  Consecutive workgroups land on consecutive XCDs (bid % 8), so with parts of a tile at consecutive bids every
// hand-over crosses XCDs - the worst case; `same_xcd` deals parts 8 apart instead.
// Three ways to make the partials visible:
//   mode 0  plain stores, __threadfence() (agent-scope release: L2 write-back) before the ticket; the reader fences
//           (acquire: invalidate) after its ticket, plain loads
//   mode 1  stores and loads with sc0 sc1 on buffer descriptors (aux bits 0 and 4: past the L2), no fence beyond the ticket's
//   mode 2  plain stores and loads, no fences: the WRONG program - how often does it fail?  (shows the probe can see the problem)
// Output per mode: wrong tiles out of N (against a host sum), us per launch, and the same launch with PARTS = 1 writing the
// tile directly (what the hand-over costs on top).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kTileFloats = 128 * 128;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_handover(const float* __restrict__ src, float* __restrict__ ws, int* __restrict__ counters,
                                                  float* __restrict__ out, int n_tiles, int parts, int same_xcd) {
    __shared__ int ticket;
    const int bid = blockIdx.x, tid = threadIdx.x;
    int tile, part;
    if (same_xcd) {                                   // parts of a tile 8 x n_tiles8 apart: the same bid % 8
        const int n8 = (n_tiles + 7) & ~7;
        tile = bid % n8;
        part = bid / n8;
        if (tile >= n_tiles) return;
    } else {
        tile = bid / parts;
        part = bid % parts;
    }
    // this part's "accumulators": a function of (tile, part, element), so the host can check the sum
    const float* mine = src + (static_cast<size_t>(tile) * parts + part) * kTileFloats;
    float* slot = ws + (static_cast<size_t>(tile) * parts + part) * kTileFloats;
    const __amdgpu_buffer_rsrc_t r_slot = rsrc_of(slot, kTileFloats * 4);
    for (int i = tid * 4; i < kTileFloats; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(mine + i);
        if (MODE == 1) {
            decltype(__builtin_amdgcn_raw_buffer_load_b128(r_slot, 0, 0, 0)) raw;
            __builtin_memcpy(&raw, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(raw, r_slot, i * 4, 0, 17);       // sc0 | sc1
        } else {
            *reinterpret_cast<float4*>(slot + i) = v;
        }
    }
    if (MODE == 0) __threadfence();
    __syncthreads();                                  // every thread's stores are issued (and, mode 0, released)
    if (tid == 0) ticket = atomicAdd(counters + tile, 1);
    __syncthreads();
    if (ticket != parts - 1) return;
    if (MODE == 0) __threadfence();
    for (int i = tid * 4; i < kTileFloats; i += 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < parts; ++p) {             // part order: the sum must not depend on who arrived last
            const float* q = ws + (static_cast<size_t>(tile) * parts + p) * kTileFloats;
            float4 v;
            if (MODE == 1) {
                const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(q, kTileFloats * 4), i * 4, 0, 17);
                __builtin_memcpy(&v, &raw, 16);
            } else {
                v = *reinterpret_cast<const float4*>(q + i);
            }
            s.x += v.x;  s.y += v.y;  s.z += v.z;  s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + static_cast<size_t>(tile) * kTileFloats + i) = s;
    }
    if (tid == 0) counters[tile] = 0;                 // ready for the next launch
}

__global__ __launch_bounds__(256) void k_direct(const float* __restrict__ src, float* __restrict__ out, int parts) {
    const int tile = blockIdx.x, tid = threadIdx.x;
    for (int i = tid * 4; i < kTileFloats; i += 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < parts; ++p) {
            const float4 v = *reinterpret_cast<const float4*>(src + (static_cast<size_t>(tile) * parts + p) * kTileFloats + i);
            s.x += v.x;  s.y += v.y;  s.z += v.z;  s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + static_cast<size_t>(tile) * kTileFloats + i) = s;
    }
}

template <int MODE>
void run(const char* what, const float* src, float* ws, int* counters, float* out, const std::vector<float>& want, int n_tiles,
         int parts, int same_xcd) {
    const int n8 = (n_tiles + 7) & ~7;
    const int grid = same_xcd ? n8 * parts : n_tiles * parts;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> got(static_cast<size_t>(n_tiles) * kTileFloats);
    int wrong_tiles = 0;
    float ms = 0.f;
    const int reps = 20;
    for (int rep = 0; rep < reps + 2; ++rep) {
        hipMemsetAsync(out, 0xff, got.size() * 4, 0);
        hipMemsetAsync(ws, 0xff, got.size() * 4 * parts, 0);           // stale workspace lines must not look right
        if (rep == 2) hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_handover<MODE>, dim3(grid), dim3(256), 0, 0, src, ws, counters, out, n_tiles, parts, same_xcd);
    }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < n_tiles; ++t) {
        bool ok = true;
        for (int i = 0; i < kTileFloats && ok; ++i) ok = got[static_cast<size_t>(t) * kTileFloats + i] == want[static_cast<size_t>(t) * kTileFloats + i];
        wrong_tiles += !ok;
    }
    std::printf("%-58s parts %d %-9s: %4d of %d tiles wrong in the last launch, %7.1f us per launch (incl. two memsets)\n", what, parts,
                same_xcd ? "same XCD" : "across", wrong_tiles, n_tiles, ms * 1e3f / reps);
}

int main() {
    const int n_tiles = 426, max_parts = 4;                            // the second round of a 120,000-pixel map
    const size_t n = static_cast<size_t>(n_tiles) * max_parts * kTileFloats;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = static_cast<float>((i * 2654435761u >> 20) & 1023) - 512.f;      // exact sums in fp32
    float *src, *ws, *out;
    int* counters;
    hipMalloc(&src, n * 4);
    hipMalloc(&ws, n * 4);
    hipMalloc(&out, static_cast<size_t>(n_tiles) * kTileFloats * 4);
    hipMalloc(&counters, n_tiles * 4);
    hipMemset(counters, 0, n_tiles * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int parts : {2, 4}) {
        std::vector<float> want(static_cast<size_t>(n_tiles) * kTileFloats);
        for (int t = 0; t < n_tiles; ++t)
            for (int i = 0; i < kTileFloats; ++i) {
                float s = 0.f;
                for (int p = 0; p < parts; ++p) s += h[(static_cast<size_t>(t) * parts + p) * kTileFloats + i];
                want[static_cast<size_t>(t) * kTileFloats + i] = s;
            }
        {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 22; ++rep) {
                if (rep == 2) hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_direct, dim3(n_tiles), dim3(256), 0, 0, src, out, parts);
            }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            std::printf("one workgroup per tile sums its %d parts from the source and writes the tile: %7.1f us per launch\n", parts, ms * 1e3f / 20);
        }
        for (int same_xcd : {0, 1}) {
            run<0>("plain stores + __threadfence() both sides", src, ws, counters, out, want, n_tiles, parts, same_xcd);
            run<1>("sc0 sc1 stores and loads (buffer aux 17), no fences", src, ws, counters, out, want, n_tiles, parts, same_xcd);
            run<2>("plain stores and loads, NO fences (the wrong program)", src, ws, counters, out, want, n_tiles, parts, same_xcd);
            hipMemset(counters, 0, n_tiles * 4);                        // (a wrong run may leave tickets behind)
        }
    }
    return 0;
}
