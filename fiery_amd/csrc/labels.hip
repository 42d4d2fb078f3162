// Instance labels of the input pipeline: centre heat map, offsets to the instance centre and the future displacement of every
// instance, from a sequence of instance-id maps.
//
// Replaces `convert_instance_mask_to_center_and_offset_label` (fiery/utils/instance.py:12-77, called per sample from
// fiery/data.py's __getitem__): the reference loops over instances x frames in Python and runs half a dozen whole-map ATen
// operators per pair (mask, two masked means, two squared-distance maps, exp, maximum, three masked assignments) - ~1,000 small
// launches per sample on a 200 x 200 map with 20 instances.  Here: one launch for the centres of all instances in all frames
// (integer sums in LDS), one for all the maps.
//   centre of instance k in frame t = round(mean row, mean column) over its pixels (round half to even, as torch.round)
//   centerness[t][p] = max over the instances present in t of exp(-((xc - x)^2 + (yc - y)^2) / sigma^2)
//   offset[t][:, p]  = (xc - x, yc - y) of p's own instance, `ignore` elsewhere
//   flow[t][:, p]    = (warped centre in t + 1) - (centre in t) for pixels of instances present in t and t + 1 whose mask,
//                      warped into frame t's ego frame (the caller resamples the id maps), is not empty; `ignore` elsewhere
#include "common.h"

namespace fiery {
namespace {

constexpr int kMaxInstances = 1023;          // ids 1 .. 1023 (three LDS tables of 1024 x 3 ints would not fit more than this)

// grid (T, 2): which = 0 the id maps, 1 the warped id maps; table[t][which][id] = (sum of rows, sum of columns, pixels)
__global__ __launch_bounds__(256) void k_instance_sums(const int* __restrict__ ids, const int* __restrict__ warped, int H, int W,
                                                       int n_inst, int* __restrict__ table) {
    __shared__ int s_sum[(kMaxInstances + 1) * 3];
    const int t = blockIdx.x, which = blockIdx.y;
    for (int i = threadIdx.x; i < (n_inst + 1) * 3; i += blockDim.x) s_sum[i] = 0;
    __syncthreads();
    const int* map = (which ? warped : ids) + static_cast<long long>(t) * H * W;
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int id = map[p];
        if (id >= 1 && id <= n_inst) {
            const int y = p / W, x = p - y * W;              // row (the reference's "x"), column (its "y")
            atomicAdd(&s_sum[id * 3], y);
            atomicAdd(&s_sum[id * 3 + 1], x);
            atomicAdd(&s_sum[id * 3 + 2], 1);
        }
    }
    __syncthreads();
    int* out = table + (static_cast<long long>(t) * 2 + which) * (n_inst + 1) * 3;
    for (int i = threadIdx.x; i < (n_inst + 1) * 3; i += blockDim.x) out[i] = s_sum[i];
}

__device__ __forceinline__ float centre_of(int sum, int count) { return rintf(static_cast<float>(sum) / static_cast<float>(count)); }

// one thread = one pixel of one frame
__global__ __launch_bounds__(256) void k_instance_labels(const int* __restrict__ ids, const int* __restrict__ table, int T, int H, int W,
                                                         int n_inst, float sigma_sq, float ignore, float* __restrict__ centerness,
                                                         float* __restrict__ offset, float* __restrict__ flow) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long hw = static_cast<long long>(H) * W;
    if (i >= hw * T) return;
    const int t = static_cast<int>(i / hw), p = static_cast<int>(i - t * hw);
    const float row = static_cast<float>(p / W), col = static_cast<float>(p % W);
    const int stride = (n_inst + 1) * 3;
    const int* now = table + static_cast<long long>(t) * 2 * stride;                 // the id map's sums of this frame
    float best = 0.f;
    for (int k = 1; k <= n_inst; ++k) {
        const int cnt = now[k * 3 + 2];
        if (cnt == 0) continue;
        const float dx = centre_of(now[k * 3], cnt) - row, dy = centre_of(now[k * 3 + 1], cnt) - col;
        const float g = expf(-(dx * dx + dy * dy) / sigma_sq);
        best = fmaxf(best, g);
    }
    centerness[i] = best;
    const int id = ids[i];
    float ox = ignore, oy = ignore, fx = ignore, fy = ignore;
    if (id >= 1 && id <= n_inst) {
        const float xc = centre_of(now[id * 3], now[id * 3 + 2]), yc = centre_of(now[id * 3 + 1], now[id * 3 + 2]);
        ox = xc - row;
        oy = yc - col;
        if (t + 1 < T) {
            const int* next = table + static_cast<long long>(t + 1) * 2 * stride;
            const int* next_warped = next + stride;
            if (next[id * 3 + 2] > 0 && next_warped[id * 3 + 2] > 0) {
                fx = centre_of(next_warped[id * 3], next_warped[id * 3 + 2]) - xc;
                fy = centre_of(next_warped[id * 3 + 1], next_warped[id * 3 + 2]) - yc;
            }
        }
    }
    offset[(static_cast<long long>(t) * 2) * hw + p] = ox;
    offset[(static_cast<long long>(t) * 2 + 1) * hw + p] = oy;
    flow[(static_cast<long long>(t) * 2) * hw + p] = fx;
    flow[(static_cast<long long>(t) * 2 + 1) * hw + p] = fy;
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int64_t fiery_instance_labels_workspace_ints(int T, int n_instances) { return static_cast<int64_t>(T) * 2 * (n_instances + 1) * 3; }

extern "C" int fiery_instance_labels(const int32_t* ids, const int32_t* warped_ids, int T, int H, int W, int n_instances, float sigma,
                                     float ignore_index, float* centerness, float* offset, float* flow, int32_t* workspace,
                                     fiery_stream_t stream) {
    FIERY_REQUIRE(ids && warped_ids && centerness && offset && flow && workspace, "instance_labels: null pointer");
    FIERY_REQUIRE(T > 0 && H > 0 && W > 0 && n_instances >= 0 && n_instances <= kMaxInstances, "instance_labels: bad shape (at most %d instances)",
                  kMaxInstances);
    FIERY_REQUIRE(static_cast<long long>(H) * H * W < (1ll << 31) && static_cast<long long>(W) * H * W < (1ll << 31),
                  "instance_labels: map too large for 32-bit coordinate sums");
    FIERY_REQUIRE(sigma > 0.f, "instance_labels: sigma must be positive");
    hipLaunchKernelGGL(k_instance_sums, dim3(T, 2), dim3(256), 0, as_stream(stream), ids, warped_ids, H, W, n_instances, workspace);
    int rc = check_launch("instance_labels (sums)");
    if (rc) return rc;
    const long long total = static_cast<long long>(T) * H * W;
    hipLaunchKernelGGL(k_instance_labels, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), ids, workspace, T, H, W, n_instances,
                       sigma * sigma, ignore_index, centerness, offset, flow);
    return check_launch("instance_labels");
}
