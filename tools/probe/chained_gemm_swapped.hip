// Probe for the Bottleneck tails: two chained GEMMs with NO transpose between them.
//
//   h = relu((A1 . W1) * scale + shift)      A1 [pixels][K1], W1 [K1][32]      (the 3 x 3 32 -> 32 product of a tail)
//   O = h . W2                               W2 [32][64]                        (its 1 x 1 up-projection)
//
// The product kernel stages h through LDS to turn the first product's accumulators (a lane holds a COUT, its registers hold
// pixels) into the second product's A operand (a lane holds a PIXEL): three LDS round trips and seven barriers per tile.  Here
// the first product's MFMAs get their operands SWAPPED - weights as the A operand, pixels as B - so the accumulator block is
// h transposed: lane l holds pixel l & 31, register r holds cout c(r, hi) = 8 (r / 4) + 4 hi + r % 4 (hi = l >> 5).  That IS
// the A-operand layout of v_mfma_f32_32x32x2_f32 (lane = row, lane half = k) if step r of the second product takes the k pair
// {c(r, 0), c(r, 1)} - any order of k is a valid order - so W2's rows are packed in that order on the host
// (W2p[r][hi][n] = W2[c(r, hi)][n]) and the second product reads h straight from the accumulator registers.
//
// One wavefront per 32-pixel tile.  Built two ways: for the CPU simulator by tests/test_probe_chained_gemm.py (which checks
// the layout algebra lane by lane against numpy), and with hipcc for the MI355X (timing against the LDS form: round 5).
#include <hip/hip_runtime.h>

typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void k_chained_swapped(const float* __restrict__ A1, const float* __restrict__ W1,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ W2p, float* __restrict__ O, int K1) {
    const int lane = threadIdx.x & 63, m = lane & 31, hi = lane >> 5;
    const int pix0 = blockIdx.x * 32;
    // ---- first product, operands swapped: acc[r] = sum_k W1[k][c(r, hi)] * A1[pix0 + m][k] --------------------------------
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int j = 0; j < K1 / 2; ++j) {
        const int k = 2 * j + hi;
        const float w = W1[k * 32 + m];                                   // A operand: row m = cout m, this lane half's k
        const float a = A1[static_cast<long long>(pix0 + m) * K1 + k];    // B operand: column m = pixel m, this lane half's k
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w, a, acc, 0, 0, 0);
    }
    // ---- BatchNorm + ReLU in registers: register r of this lane half is cout c(r, hi) ------------------------------------
    float h[16];
    for (int r = 0; r < 16; ++r) {
        const int c = 8 * (r >> 2) + 4 * hi + (r & 3);
        h[r] = fmaxf(fmaf(acc[r], scale[c], shift[c]), 0.f);
    }
    // ---- second product straight from the registers: step r takes the k pair {c(r, 0), c(r, 1)} -------------------------
    v16f out[2];
    for (int nt = 0; nt < 2; ++nt)
        for (int r = 0; r < 16; ++r) out[nt][r] = 0.f;
    for (int r = 0; r < 16; ++r) {
        for (int nt = 0; nt < 2; ++nt) {
            const float b = W2p[(r * 2 + hi) * 64 + nt * 32 + m];         // B operand: k = this lane half's cout of step r
            out[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[r], b, out[nt], 0, 0, 0);
        }
    }
    // ---- the result in the standard layout: lane = cout, registers = pixels ----------------------------------------------
    for (int nt = 0; nt < 2; ++nt)
        for (int r = 0; r < 16; ++r) {
            const int pl = 8 * (r >> 2) + 4 * hi + (r & 3);
            O[static_cast<long long>(pix0 + pl) * 64 + nt * 32 + m] = out[nt][r];
        }
}

// pointers as the device sees them (simulator: host memory); n_pixels a multiple of 32, K1 even
extern "C" int probe_chained_swapped(const float* A1, const float* W1, const float* scale, const float* shift, const float* W2p,
                                     float* O, int n_pixels, int K1) {
    if (n_pixels % 32 != 0 || K1 % 2 != 0 || n_pixels <= 0) return 1;
    hipLaunchKernelGGL(k_chained_swapped, dim3(n_pixels / 32), dim3(64), 0, 0, A1, W1, scale, shift, W2p, O, K1);
    return hipGetLastError() == hipSuccess ? 0 : 2;                       // (default stream; on the GPU the caller synchronises)
}
