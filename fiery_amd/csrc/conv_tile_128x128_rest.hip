// The 128 x 128 tile's generic and small-cin kernels (see conv_tile_128x128.hip).
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_128x128_rest(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk) {
    return conv_launch_tile<128, 128, 5u>(p, grid, stream, variant, clk);
}
}  // namespace fiery
