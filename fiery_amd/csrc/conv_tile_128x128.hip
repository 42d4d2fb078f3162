// The 128 x 128 tile of the implicit-GEMM convolution (conv_igemm_kernel.h): the scalar-addressed kernel here, the
// generic and small-cin ones in conv_tile_128x128_rest.hip (each takes minutes to compile).
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_128x128(const ConvP& p, dim3 grid, hipStream_t stream, int variant, unsigned long long* clk) {
    if (variant == kConvGeneric || variant == kConvSmallCin) return conv_launch_128x128_rest(p, grid, stream, variant, clk);
    return conv_launch_tile<128, 128, 2u>(p, grid, stream, variant, clk);
}
}  // namespace fiery
