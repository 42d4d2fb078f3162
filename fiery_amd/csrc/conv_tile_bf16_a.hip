// bf16 matrix-core form of the implicit-GEMM convolution (conv_igemm_kernel.h), 128-pixel tiles; a translation unit of
// its own so that it compiles beside the fp32 tiles.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_bf16_m128(const ConvP& p, int bn, dim3 grid, hipStream_t stream) {
    if (bn == 32) conv_launch_tile_bf16<128, 32>(p, grid, stream);
    else if (bn == 64) conv_launch_tile_bf16<128, 64>(p, grid, stream);
    else if (bn == 128) conv_launch_tile_bf16<128, 128>(p, grid, stream);
    else return false;
    return true;
}
bool conv_launch_bf16_halo_m128(const ConvP& p, int bn, dim3 grid, hipStream_t stream) {
    if (bn == 64) conv_launch_tile_bf16_halo<128, 64>(p, grid, stream);
    else if (bn == 128) conv_launch_tile_bf16_halo<128, 128>(p, grid, stream);
    else return false;
    return true;
}
}  // namespace fiery
