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

// min(max(a - b, 0), 1) on both halves in one instruction (neg modifier on the second source + clamp)
__device__ __forceinline__ v2f pk_sub_sat(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Reduce-scatter of three values over the wavefront's four 16-lane rows: every lane passes its (a, b, c); afterwards a lane
// of row 0 holds the four rows' sum of a, row 1 that of b, row 2 that of c (row 3: unspecified).  gfx950's row swaps do a
// reduce-scatter step in one instruction: v_permlane16_swap(x, y) leaves even rows with (own x, neighbour's x) and odd
// rows with (neighbour's y, own y), so one add gives even rows the pair sum of x and odd rows that of y; v_permlane32_swap
// does the same for the two halves.  Three swaps and three adds.
__device__ __forceinline__ float rows_reduce_scatter3(float a, float b, float c) {
    const auto ab = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float p = __uint_as_float(ab[0]) + __uint_as_float(ab[1]);           // even rows: a (pair sum), odd rows: b
    const auto cz = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), 0u, false, false);
    const float t = __uint_as_float(cz[0]) + __uint_as_float(cz[1]);           // even rows: c (pair sum), odd rows: junk
    const auto pt = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(t), false, false);
    return __uint_as_float(pt[0]) + __uint_as_float(pt[1]);                    // rows 0, 1: p of both halves; rows 2, 3: t
}

// Sum of a value over the wavefront's four 16-lane rows (lanes l, l ^ 16, l ^ 32, l ^ 48), the same total in all four:
// gfx950's row swaps exchange odd and even rows (v_permlane16_swap) and the two halves (v_permlane32_swap) of a register
// pair in one vector-ALU instruction each - with both operands holding the value, the two results are "my row" and "the
// other row" in every lane.
__device__ __forceinline__ float rows_sum4(float v) {
    const unsigned bits = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
    const float pair = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned pbits = __float_as_uint(pair);
    const auto b = __builtin_amdgcn_permlane32_swap(pbits, pbits, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- bf16 matrix-core operands ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float fiery_v16f __attribute__((ext_vector_type(16)));

// eight fp32 -> eight bf16, round to nearest even (four v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x8 pack_bf16x8(float4 lo, float4 hi) {
    bf16x8 v;
    v[0] = static_cast<__bf16>(lo.x);  v[1] = static_cast<__bf16>(lo.y);  v[2] = static_cast<__bf16>(lo.z);  v[3] = static_cast<__bf16>(lo.w);
    v[4] = static_cast<__bf16>(hi.x);  v[5] = static_cast<__bf16>(hi.y);  v[6] = static_cast<__bf16>(hi.z);  v[7] = static_cast<__bf16>(hi.w);
    return v;
}
__device__ __forceinline__ bf16x8 load_bf16x8(const float* p) { return *reinterpret_cast<const bf16x8*>(p); }
// sixteen bytes that already hold eight bf16 values (a piece of the packed weights fetched into registers)
__device__ __forceinline__ bf16x8 bits_bf16x8(float4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
// four fp32 -> four bf16 (two v_cvt_pk_bf16_f32), as the eight bytes they occupy in memory
__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h;
    h[0] = static_cast<__bf16>(v.x);  h[1] = static_cast<__bf16>(v.y);  h[2] = static_cast<__bf16>(v.z);  h[3] = static_cast<__bf16>(v.w);
    uint2 r;
    __builtin_memcpy(&r, &h, 8);
    return r;
}
// four bf16 (as packed by pack_bf16x4) -> four fp32, exact
__device__ __forceinline__ float4 unpack_bf16x4(uint2 h) {
    return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
}
// D(32 x 32) += A(32 x 16) . B(16 x 32): lane l holds A[l & 31][8 (l >> 5) + j], B[8 (l >> 5) + j][l & 31], j < 8
__device__ __forceinline__ fiery_v16f mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, fiery_v16f c) {
#ifdef FIERY_BF16_REPEAT                   // timing experiment (wrong results): every bf16 MFMA issued this many times
#pragma unroll
    for (int i = 1; i < FIERY_BF16_REPEAT; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// the nearest bf16 value (round to nearest even) as a float
__device__ __forceinline__ float bf16_round(float v) { return static_cast<float>(static_cast<__bf16>(v)); }
// Eight fp32 values as three bf16 terms each: x = t1 + t2 + t3 exactly (3 x 8 significand bits; the remainders x - t1 and
// (x - t1) - t2 are exact in fp32).  With the partial products of weight >= 2^-24 a product of two such values on the bf16
// matrix cores is as accurate as the fp32 matrix instruction's (conv_winograd.hip, split form).
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& t1, bf16x8& t2, bf16x8& t3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = static_cast<__bf16>(x[j]);
        const float r1 = x[j] - static_cast<float>(a);
        const __bf16 b = static_cast<__bf16>(r1);
        const float r2 = r1 - static_cast<float>(b);
        t1[j] = a;
        t2[j] = b;
        t3[j] = static_cast<__bf16>(r2);
    }
}
// one fp32 -> bf16 bits (round to nearest even), for the weight packer
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    const __bf16 b = static_cast<__bf16>(v);
    unsigned short r;
    __builtin_memcpy(&r, &b, 2);
    return r;
}

// Code-motion fence: everything that produces a, b, c is issued before, every memory access written after it
// stays after.  (The compiler otherwise sinks a batch's arithmetic below the next batch's loads and then holds
// three batches of rows in registers - or rather in scratch.)
__device__ __forceinline__ void pk_pin(v2f& a, v2f& b, v2f& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }


// Wavefront-level rendezvous for data handed from lane to lane through LDS inside ONE wavefront: the LDS executes a
// wavefront's instructions in order, so no s_barrier is needed - only that the compiler keeps the accesses in program order
// (fence at wavefront scope) and the wavefront's lanes are treated as having arrived (wave_barrier: no instruction).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Every vector-memory access this wavefront has issued has completed (loads returned, stores acknowledged): in front of the
// barrier + ticket that hands sc0 sc1 (written-through) partial results to another workgroup.
__device__ __forceinline__ void vmem_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// After a 16-byte buffer store whose data registers the next vector instruction overwrites: the store reads its data some
// cycles after issue, and a packed-fp32 write (v_pk_fma_f32) of those registers right behind it was seen to reach memory in
// their place (round 5, conv_winograd.hip: the second and third pixel of a block swapped in some lanes) - the compiler's
// hazard recogniser does not cover this pair, so the stores are followed by explicit wait states.
// (sched_barrier on both sides: an asm statement alone does not keep vector arithmetic from being scheduled across it)
__device__ __forceinline__ void store_data_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// A value the optimiser must take as it finds it at this point (per-lane / wave-uniform): what is derived from it inside a
// loop body is recomputed there instead of being hoisted out and kept in registers across the body.
__device__ __forceinline__ void opaque_v(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque_s(int& v) { asm volatile("" : "+s"(v)); }

// The kernel's argument block (its first and only explicit argument), re-read from the kernarg segment behind an offset the
// optimiser cannot see through: inside a loop the loads stay inside the loop (scalar loads from constant memory, cached)
// instead of being hoisted and kept live across the body.
template <typename Args>
__device__ __forceinline__ const Args& kernel_args_again(const Args&) {
    int off = 0;
    asm volatile("" : "+s"(off));
    const __attribute__((address_space(4))) char* ka =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const Args*)(ka + off);
}

}  // namespace fiery
