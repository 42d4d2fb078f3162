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

}  // namespace fiery
