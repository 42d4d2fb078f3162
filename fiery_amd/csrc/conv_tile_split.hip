// Split form of the implicit-GEMM convolution (conv_igemm_kernel.h, SPLIT): fp32 accuracy on the bf16 matrix cores - operands
// as three bf16 terms, six partial products per product.  128-pixel tiles, 32 couts (the chained Bottleneck tails too) or 64.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_split(const ConvP& p, int bn, dim3 grid, hipStream_t stream) {
    if (bn == 32) conv_launch_tile_split<128, 32>(p, grid, stream);
    else if (bn == 64) conv_launch_tile_split<128, 64>(p, grid, stream);
    else return false;
    return true;
}
}  // namespace fiery
