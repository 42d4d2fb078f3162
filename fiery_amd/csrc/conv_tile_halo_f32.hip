// fp32 halo loop of the implicit-GEMM convolution (conv_igemm_kernel.h): 3 x 3 / stride 1 layers on 64-pixel tiles.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_f32_halo(const ConvP& p, int bn, dim3 grid, hipStream_t stream) {
    if (bn == 64) conv_launch_tile_f32_halo<64, 64>(p, grid, stream);
    else if (bn == 128) conv_launch_tile_f32_halo<64, 128>(p, grid, stream);
    else return false;
    return true;
}
}  // namespace fiery
