// Packed-fp32 helpers for gfx950: two floats per lane in a 64-bit register pair, one VALU issue for both
// (v_pk_add_f32 / v_pk_fma_f32 run at the full vector rate, so packed code needs half the issue slots).
// Included with angle brackets: the CPU simulator of tests/sim/ supplies a plain C++ stand-in of the same name.
#pragma once
#include <hip/hip_runtime.h>

namespace fiery {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_make(float lo, float hi) { return v2f{lo, hi}; }
__device__ __forceinline__ v2f pk_splat(float v) { return v2f{v, v}; }
__device__ __forceinline__ float pk_lo(v2f v) { return v.x; }
__device__ __forceinline__ float pk_hi(v2f v) { return v.y; }

// a * b + c, both halves, one rounding each
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// min(max(a + b, 0), 1) on both halves in ONE instruction: the VOP3P clamp bit.  The compiler has no pattern that
// folds a clamp into a packed add (it emits one v_max_f32 ... clamp per half), hence the asm.  `b` is wave-uniform
// (a scalar register pair).
__device__ __forceinline__ v2f pk_add_sat_uniform(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

// Code-motion fence: everything that produces a, b, c is issued before, every memory access written after it
// stays after.  (The compiler otherwise sinks a batch's arithmetic below the next batch's loads and then holds
// three batches of rows in registers - or rather in scratch.)
__device__ __forceinline__ void pk_pin(v2f& a, v2f& b, v2f& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }

}  // namespace fiery
