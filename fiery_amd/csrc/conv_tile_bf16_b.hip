// bf16 matrix-core form of the implicit-GEMM convolution (conv_igemm_kernel.h), 64-pixel tiles, and the dispatcher.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_bf16_m128(const ConvP& p, int bn, dim3 grid, hipStream_t stream);
bool conv_launch_bf16_halo_m128(const ConvP& p, int bn, dim3 grid, hipStream_t stream);

bool conv_launch_bf16(const ConvP& p, int bm, int bn, dim3 grid, hipStream_t stream, bool halo) {
    if (bm == 128) return halo ? conv_launch_bf16_halo_m128(p, bn, grid, stream) : conv_launch_bf16_m128(p, bn, grid, stream);
    if (bm != 64) return false;
    if (halo) {                                  // 3 x 3, stride 1: the A tile and its neighbours fetched once per channel group
        if (bn == 64) conv_launch_tile_bf16_halo<64, 64>(p, grid, stream);
        else if (bn == 128) conv_launch_tile_bf16_halo<64, 128>(p, grid, stream);
        else return false;
        return true;
    }
    if (bn == 64) conv_launch_tile_bf16<64, 64>(p, grid, stream);
    else if (bn == 128) conv_launch_tile_bf16<64, 128>(p, grid, stream);
    else return false;
    return true;
}
}  // namespace fiery
