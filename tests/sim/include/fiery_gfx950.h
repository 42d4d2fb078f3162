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

}  // namespace fiery
