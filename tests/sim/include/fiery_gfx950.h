// TEST INFRASTRUCTURE: CPU stand-in for fiery_amd/csrc/fiery_gfx950.h (packed fp32 helpers), same semantics.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace fiery {

struct v2f {
    float x, y;
};
inline v2f operator+(v2f a, v2f b) { return {a.x + b.x, a.y + b.y}; }
inline v2f operator-(v2f a, v2f b) { return {a.x - b.x, a.y - b.y}; }
inline v2f& operator+=(v2f& a, v2f b) { a = a + b; return a; }
inline v2f pk_make(float lo, float hi) { return {lo, hi}; }
inline v2f pk_splat(float v) { return {v, v}; }
inline float pk_lo(v2f v) { return v.x; }
inline float pk_hi(v2f v) { return v.y; }
inline v2f pk_fma(v2f a, v2f b, v2f c) { return {std::fmaf(a.x, b.x, c.x), std::fmaf(a.y, b.y, c.y)}; }
inline float hipsim_sat(float v) { return v != v ? 0.f : (v < 0.f ? 0.f : (v > 1.f ? 1.f : v)); }
inline v2f pk_add_sat_uniform(v2f a, v2f b) { return {hipsim_sat(a.x + b.x), hipsim_sat(a.y + b.y)}; }

inline void pk_pin(v2f&, v2f&, v2f&) {}

// ---- bf16 matrix-core operands: eight bf16 values as their bit patterns ---------------------------------------------
struct bf16x8 {
    unsigned short v[8];
};
typedef hipsim_v16f fiery_v16f;
inline unsigned short bf16_bits(float f) {                               // round to nearest even, NaN kept quiet
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
}
inline float bf16_value(unsigned short b) {
    const unsigned u = static_cast<unsigned>(b) << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
inline float bf16_round(float v) { return bf16_value(bf16_bits(v)); }
inline void split_bf16x8(const float (&x)[8], bf16x8& t1, bf16x8& t2, bf16x8& t3) {
    for (int j = 0; j < 8; ++j) {
        const float a = bf16_round(x[j]), r1 = x[j] - a, b = bf16_round(r1), r2 = r1 - b;
        t1.v[j] = bf16_bits(a);
        t2.v[j] = bf16_bits(b);
        t3.v[j] = bf16_bits(r2);
    }
}
inline bf16x8 pack_bf16x8(float4 lo, float4 hi) {
    return {{bf16_bits(lo.x), bf16_bits(lo.y), bf16_bits(lo.z), bf16_bits(lo.w), bf16_bits(hi.x), bf16_bits(hi.y), bf16_bits(hi.z),
             bf16_bits(hi.w)}};
}
inline uint2 pack_bf16x4(float4 v) {
    const unsigned short h[4] = {bf16_bits(v.x), bf16_bits(v.y), bf16_bits(v.z), bf16_bits(v.w)};
    uint2 r;
    __builtin_memcpy(&r, h, 8);
    return r;
}
inline float4 unpack_bf16x4(uint2 h) {
    unsigned short b[4];
    __builtin_memcpy(b, &h, 8);
    return make_float4(bf16_value(b[0]), bf16_value(b[1]), bf16_value(b[2]), bf16_value(b[3]));
}
inline bf16x8 bits_bf16x8(float4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
inline bf16x8 load_bf16x8(const float* p) {
    bf16x8 r;
    __builtin_memcpy(&r, p, 16);
    return r;
}
// D(32 x 32) += A(32 x 16) . B(16 x 32), one wave: lane l holds A[l & 31][8 (l >> 5) + j], B[8 (l >> 5) + j][l & 31]; D as the
// fp32 32 x 32 MFMA (row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = l & 31).  Products of bf16 values are exact in fp32.
inline fiery_v16f mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, fiery_v16f c) {
    ::hipsim::Run& run = *::hipsim::run_ptr();
    const ::hipsim::Tls& t = ::hipsim::tls();
    const int l = t.lane, w = t.wave;
    float* wa = run.wave_a[w].data();
    float* wb = run.wave_b[w].data();
    for (int j = 0; j < 8; ++j) {
        wa[l * 8 + j] = bf16_value(a.v[j]);
        wb[l * 8 + j] = bf16_value(b.v[j]);
    }
    ::hipsim::barrier_wait(run.waves[w]);
    const int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const int src_a = (row + 32 * (k >> 3)) * 8 + (k & 7);       // lane (row, k / 8) holds A[row][k]
            const int src_b = (col + 32 * (k >> 3)) * 8 + (k & 7);
            acc = std::fmaf(wa[src_a], wb[src_b], acc);
        }
        c[r] = acc;
    }
    ::hipsim::barrier_wait(run.waves[w]);
    return c;
}

inline v2f pk_sub_sat(v2f a, v2f b) { return {hipsim_sat(a.x - b.x), hipsim_sat(a.y - b.y)}; }

// reduce-scatter of (a, b, c) over the four 16-lane rows: row 0 gets sum(a), row 1 sum(b), row 2 sum(c); same order of
// additions as the hardware form (pair sums of neighbouring rows first, then the two halves, lower half first)
inline float rows_reduce_scatter3(float a, float b, float c) {
    const int lane = ::hipsim::tls().lane, row = lane >> 4;
    const float pa = __shfl_xor(a, 16), pb = __shfl_xor(b, 16), pc = __shfl_xor(c, 16);
    // even rows: own a + neighbour's a (own first); odd rows: neighbour's b + own b
    const float p = (row & 1) ? pb + b : a + pa;
    const float t = (row & 1) ? 0.f + 0.f : c + pc;
    const float fp = __shfl_xor(p, 32), ft = __shfl_xor(t, 32);
    // lower half: own p + far p; upper half: far t + own t
    return (row >> 1) ? ft + t : p + fp;
}

// sum over the wavefront's four 16-lane rows, in the order the hardware form adds them: (row ^ 1 pair) + (other pair)
inline float rows_sum4(float v) {
    const float other = __shfl_xor(v, 16);
    const int lane = ::hipsim::tls().lane;
    const float pair = ((lane >> 4) & 1) ? other + v : v + other;        // even row first, as a[0] + a[1]
    const float far = __shfl_xor(pair, 32);
    return (lane >> 5) ? far + pair : pair + far;                           // lower half first, as b[0] + b[1]
}


// wavefront-level rendezvous (lane-to-lane hand-over through LDS inside one wavefront)
inline void wave_sync() { ::hipsim::barrier_wait(::hipsim::run_ptr()->waves[::hipsim::tls().wave]); }

inline void vmem_done() {}
inline void store_data_settle() {}

inline void opaque_v(int&) {}
inline void opaque_s(int&) {}
// (simulator: the argument block is the argument)
template <typename Args>
inline const Args& kernel_args_again(const Args& a) { return a; }

}  // namespace fiery
