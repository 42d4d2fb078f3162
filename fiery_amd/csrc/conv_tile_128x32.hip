// The 128 x 32 tile of the implicit-GEMM convolution (conv_igemm_kernel.h), a translation unit of its own so that
// the tile shapes compile in parallel.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_128x32(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk) {
    return conv_launch_tile<128, 32, 7u>(p, grid, stream, variant, clk);      // variant mask: see ConvVariant
}
}  // namespace fiery
