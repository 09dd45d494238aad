// Carry-free Montgomery arithmetic for the Pasta fields on gfx950: nine SIGNED 29-bit limbs, R9 = 2^261.
//
// Why a second representation next to field.cuh (8 x 32-bit limbs, R = 2^256 -- the reference's memory layout):
// on this chip a carry-writing VALU add costs about as much as a multiply-add, and the 8 x 32 multiplier spends
// 96 of its 248 issue slots on them.  With 29-bit limbs a whole column of partial products
//     sum_{i+j=k} a_i b_j  +  sum_i m_i p_{k-i}            (<= 9 + 5 terms of < 2^60)
// fits a 64-bit accumulator, so one partial product is ONE v_mad_i64_i32 and a column ends with a shift and a mask:
// 81 + 45 + 9 multiply-adds and ~45 plain VALU ops per modular multiplication, no carry chain anywhere.
//
// Limbs are signed (int32): a - b is nine v_sub_u32, no bias, and values may be negative.  Bounds ("mul-ready"):
//   * limbs of both operands of a multiplication satisfy 9 |a_i| |b_j| + 2^61 < 2^63, e.g. both below 2^30.4, or one
//     below 2^29 ("normalised": limbs 0..7 in [0, 2^29), limb 8 signed) and the other below 2^31.8;
//   * |value| < 2^258 for both operands.  Then the product is normalised and lies in (-2^255, 2^255 + p): R9 / p = 2^7, so
//     there is never a conditional subtraction and sums / differences of a few products are again valid operands.
// fe9_norm is the carry pass (3 ops per limb) that turns any value with limbs below 2^31 into a normalised one.
//
// Memory formats never change: a field element in HBM is 8 x u32.  fe9_unpack / fe9_pack convert at load / store.
// "M9 form" of x is x * 2^261 mod p; the reference's Montgomery form is x * 2^256 mod p (fe9_from_r256 / fe9_to_r256).
#pragma once
#include "field.cuh"

namespace h2 {

typedef int32_t i32;
typedef int64_t i64;

struct fe9 {
    i32 v[9];
};
static constexpr u32 M29 = (1u << 29) - 1;
// constants computed by gen_field9_consts.py (checked on the device by tests/native/field_check.hip):
//   Mod9<F>::P1..P4 modulus limbs;  fe9_one = 2^261 mod p (1 in M9 form);  fe9_r2 = 2^522 mod p;
//   fe9_k_in  = 2^266 mod p : fe9_mul(unpack(x * 2^256), k_in) = x * 2^261   (reference Montgomery form -> M9 form)
//   fe9_k_out = 2^256 mod p : fe9_mul(x * 2^261, k_out)        = x * 2^256   (M9 form -> reference Montgomery form)
//   fe9_p16 = 16 p, fe9_p_shl(sh) = p << sh in normalised limbs
#include "field9_consts.inc"

// p in 29-bit limbs: [1, p1, p2, p3, p4, 0, 0, 0, 2^22]   (p = 2^254 + t, t < 2^126); Mod9<F>::P1..P4 are generated
static constexpr i32 P9_8 = 1 << 22;

__device__ __forceinline__ fe9 fe9_zero() { return fe9{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }

__device__ __forceinline__ fe9 fe9_add(const fe9 &a, const fe9 &b) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
__device__ __forceinline__ fe9 fe9_sub(const fe9 &a, const fe9 &b) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
__device__ __forceinline__ fe9 fe9_dbl(const fe9 &a) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
    return r;
}
// carry pass: limbs 0..7 into [0, 2^29), limb 8 keeps the sign
__device__ __forceinline__ fe9 fe9_norm(const fe9 &a) {
    fe9 r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const i32 t = a.v[i] + c;
        r.v[i] = t & (i32)M29;
        c = t >> 29;
    }
    r.v[8] = a.v[8] + c;
    return r;
}

// 4 a, normalised, for a PRODUCT a (limbs 0..7 in [0, 2^29], limb 0 may equal 2^29: 4 * 2^29 does not fit an i32, so the low
// limbs are shifted and carried as unsigned words; limb 8 is small and signed)
__device__ __forceinline__ fe9 fe9_quadruple_norm(const fe9 &a) {
    fe9 r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 t = ((u32)a.v[i] << 2) + c;
        r.v[i] = (i32)(t & M29);
        c = t >> 29;
    }
    r.v[8] = a.v[8] * 4 + (i32)c;
    return r;
}

// opaque constants: keep hipcc from turning "* 1" / "* 2^22" into 64-bit shift-and-add sequences (3 carry-linked
// instructions instead of one multiply-add)
__device__ __forceinline__ i32 opaque(i32 x) {
    asm volatile("" : "+s"(x));
    return x;
}

// Montgomery product a * b * 2^-261 mod p, product scanning with one signed 64-bit accumulator.
template <int F, bool SQUARE> __device__ __forceinline__ fe9 fe9_mul_impl(const fe9 &a, const fe9 &b) {
    i64 acc = 0;
    i32 m[9];
    fe9 r;
    i32 a2[9];
    if (SQUARE) {
#pragma unroll
        for (int i = 0; i < 9; i++) a2[i] = a.v[i] << 1;
    }
    const i32 c22 = opaque(P9_8), one = opaque(1);
#pragma unroll
    for (int k = 0; k < 17; k++) {
        if (SQUARE) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j > i && j < 9) acc += (i64)a.v[i] * a2[j];
            }
            if (!(k & 1)) acc += (i64)a.v[k / 2] * a.v[k / 2];
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) acc += (i64)a.v[i] * b.v[j];
            }
        }
        if (k >= 1 && k - 1 < 9) acc += (i64)m[k - 1] * Mod9<F>::P1;
        if (k >= 2 && k - 2 < 9) acc += (i64)m[k - 2] * Mod9<F>::P2;
        if (k >= 3 && k - 3 < 9) acc += (i64)m[k - 3] * Mod9<F>::P3;
        if (k >= 4 && k - 4 < 9) acc += (i64)m[k - 4] * Mod9<F>::P4;
        if (k >= 8 && k - 8 < 9) acc += (i64)m[k - 8] * c22;
        if (k < 9) {
            m[k] = (i32)((0u - (u32)acc) & M29);     // -p^-1 = -1 mod 2^29
            acc += (i64)m[k] * one;                  // p0 = 1: the low 29 bits cancel
        } else {
            r.v[k - 9] = (i32)((u32)acc & M29);
        }
        acc >>= 29;
    }
    r.v[8] = (i32)acc;
    return r;
}
// the same in plain C (the compiler picks the instructions): readable specification and A/B baseline
template <int F> __device__ __forceinline__ fe9 fe9_mul_c(const fe9 &a, const fe9 &b) { return fe9_mul_impl<F, false>(a, b); }
template <int F> __device__ __forceinline__ fe9 fe9_sqr_c(const fe9 &a) { return fe9_mul_impl<F, true>(a, a); }

// The shipped multiplier / squarer: one asm statement each, generated by gen_field9_mul.py (126 / 90 multiply-adds, 16 shifts + one funnel shift,
// 17 bit ops; limb 0 of the result lies in [1, 2^29], see the generator's docstring).
#ifndef H2_FE9_IMPL
#define H2_FE9_IMPL 1
#endif
template <int F> __device__ __forceinline__ fe9 fe9_mul(const fe9 &a, const fe9 &b) {
#if H2_FE9_IMPL == 0
    return fe9_mul_c<F>(a, b);
#else
#include "field9_mul.inc"
    return r;
#endif
}
template <int F> __device__ __forceinline__ fe9 fe9_sqr(const fe9 &a) {
#if H2_FE9_IMPL == 0
    return fe9_sqr_c<F>(a);
#else
#include "field9_sqr.inc"
    return r;
#endif
}

// ---- fused forms for the point formulas (generated, gen_field9_mul.py) ----------------------------------------------------------
// fe9_dot2(a, b, c, d) = (a b + c d) 2^-261: both products share one pass over the columns and ONE Montgomery reduction (207
// multiply-adds against 2 x 126, and the difference of two products needs no subtraction / carry pass afterwards).  A column holds
// at most 18 + 5 terms below 2^58: limbs of all four operands in [-2^29, 2^29], |a b + c d| < 2^516 for a normalised result.
// fe9_sqr_minus(a, s) = a^2 2^-261 - s for a signed limb vector s (limbs below 2^31 in magnitude): s_i enters column 9 + i with
// weight -1, the result leaves normalised like any product (value: that of the square, in (-2^255, 2^255 + p), minus s).
template <int F> __device__ __forceinline__ fe9 fe9_dot2(const fe9 &a, const fe9 &b, const fe9 &c, const fe9 &d) {
#if H2_FE9_IMPL == 0
    return fe9_norm(fe9_add(fe9_mul_c<F>(a, b), fe9_mul_c<F>(c, d)));
#else
#include "field9_dot2.inc"
    return r;
#endif
}
template <int F> __device__ __forceinline__ fe9 fe9_sqr_minus(const fe9 &a, const fe9 &s) {
#if H2_FE9_IMPL == 0
    return fe9_norm(fe9_sub(fe9_sqr_c<F>(a), s));
#else
#include "field9_sqr_minus.inc"
    return r;
#endif
}

// ---- 8 x 32 <-> 9 x 29 repacking (values, not Montgomery forms) -------------------------------------------------------
__device__ __forceinline__ fe9 fe9_unpack(const fe &a) {
    fe9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        u32 lo = a.v[w] >> s;
        if (s > 3 && w + 1 < 8) lo |= a.v[w + 1] << (32 - s);
        r.v[i] = (i32)(lo & M29);
    }
    return r;
}
// normalised, non-negative, < 2^256
__device__ __forceinline__ fe fe9_pack(const fe9 &a) {
    fe r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // word w holds bits [32w, 32w + 32): limbs i with 29 i < 32 w + 32 and 29 i + 29 > 32 w
        u32 x = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int lo = 29 * i - 32 * w;     // position of limb i's bit 0 inside word w
            if (lo > -29 && lo < 32) x |= lo >= 0 ? (u32)a.v[i] << lo : (u32)a.v[i] >> (-lo);
        }
        r.v[w] = x;
    }
    return r;
}

// canonical representative in [0, p) of any valid operand (|value| < 2^258), packed 8 x 32
template <int F> __device__ __forceinline__ fe fe9_canonical(const fe9 &a) {
    // a + 16 p > 0; 16 p = 2^258 + 16 t: add it limb-wise (its 29-bit limbs), carry, then reduce below p
    fe9 t = a;
    const fe9 p16 = fe9_p16<F>();
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] += p16.v[i];
    t = fe9_norm(t);                            // value in (0, 2^259)
    // rare (flushes, equality tests): conditional subtraction of 16p, 8p, 4p, 2p, p, each a limb-wise subtract + carry pass
#pragma unroll
    for (int sh = 4; sh >= 0; sh--) {
        fe9 d;
        const fe9 pk = fe9_p_shl<F>(sh);
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] = t.v[i] - pk.v[i];
        d = fe9_norm(d);
        if (d.v[8] >= 0) t = d;
    }
    return fe9_pack(t);
}

// the same for a value known to lie in (-p, 2p): what a multiplication by a CANONICAL constant returns (|a| < 2^258 times
// k < p over 2^261 is below 2^252 in magnitude, plus the quotient's p).  One shift by p and two conditional subtractions.
template <int F> __device__ __forceinline__ fe fe9_canonical_small(const fe9 &a) {
    const fe9 pk = fe9_p_shl<F>(0);
    fe9 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] = a.v[i] + pk.v[i];
    t = fe9_norm(t);                            // (0, 3p)
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        fe9 d;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] = t.v[i] - pk.v[i];
        d = fe9_norm(d);
        if (d.v[8] >= 0) t = d;
    }
    return fe9_pack(t);
}

template <int F> __device__ __forceinline__ fe9 fe9_from_r256(const fe &x_r256) { return fe9_mul<F>(fe9_unpack(x_r256), fe9_k_in<F>()); }
template <int F> __device__ __forceinline__ fe fe9_to_r256(const fe9 &x_m9) { return fe9_canonical_small<F>(fe9_mul<F>(x_m9, fe9_k_out<F>())); }

// value = 0 mod p ?  a: limbs below 2^31, |value| < 2^258.  p = 1 mod 2^29, so k p = k mod 2^29: unless the low limb is
// within +-16 of a multiple of 2^29 the answer is no (the common case costs three instructions).
__device__ __forceinline__ bool fe9_maybe_zero_mod_p(const fe9 &a) { return (((u32)a.v[0] + 16u) & M29) <= 32u; }
template <int F> __device__ __forceinline__ bool fe9_is_zero_mod_p(const fe9 &a) {
    if (!fe9_maybe_zero_mod_p(a)) return false;
    return fe_is_zero(fe9_canonical<F>(a));
}

}  // namespace h2
