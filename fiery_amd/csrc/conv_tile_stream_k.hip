// Stream-K form of the implicit-GEMM convolution (conv_igemm_kernel.h, round 5): 128-pixel tiles, 64 or 128 couts wide, the
// scalar-addressed fp32 loop; a translation unit of its own so that it compiles beside the other tile shapes.
#define FIERY_CONV_KERNEL_TU 1
#include "conv_igemm_kernel.h"

namespace fiery {
bool conv_launch_stream_k(const ConvP& p, int bn, dim3 grid, hipStream_t stream) {
    if (bn == 128) conv_launch_tile_stream_k<128, 128>(p, grid, stream);
    else if (bn == 64) conv_launch_tile_stream_k<128, 64>(p, grid, stream);
    else return false;
    return true;
}
int conv_stream_k_per_cu(int bn) { return bn == 128 ? conv_waves_per_simd(128, 128, true) : conv_waves_per_simd(128, 64, true); }
}  // namespace fiery
