// 255-bit Montgomery arithmetic for the Pasta fields on gfx950 (CDNA4), register-resident.
//
// Replaces the L0 field ops the reference calls from its hot loops (pasta_curves 0.5.1 Fp/Fq,
// used at halo2_proofs/src/arithmetic.rs:242-246,288-292 and inside every point addition).
// Representation: 8 x 32-bit little-endian limbs, Montgomery form with R = 2^256 -- bit-identical
// to the 4 x u64 limbs a Rust `Fp`/`Fq` holds in memory, so buffers cross the FFI without conversion.
//
// Both moduli are 2^254 + t with t < 2^126:  32-bit limbs  [1, p1, p2, p3, 0, 0, 0, 2^30]  and
// -p^-1 mod 2^32 = 0xffffffff, so the Montgomery quotient digit is a negation and the reduction
// needs 4 multiplies per digit instead of 8.
//
// The multiplier is a product-scanning (Comba) Montgomery multiply: every column keeps a 96-bit
// accumulator {lo:mid (aligned VGPR pair), hi}; one partial product costs
//     v_mad_u64_u32 acc, vcc, a, b, acc ; v_addc_co_u32 hi, vcc, 0, hi, vcc
// i.e. 96 quarter-rate multiplies + 96 carry adds per modular multiplication.  No MFMA: this is
// modular-integer work, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_inv.cuh"

namespace h2 {

typedef uint32_t u32;
typedef uint64_t u64;

enum { FP = 0, FQ = 1 };

struct fe {
    u32 v[8];
};

template <int F> struct Mod;
template <> struct Mod<FP> {  // Pallas base / Vesta scalar
    static constexpr u32 P1 = 0x992d30edu, P2 = 0x094cf91bu, P3 = 0x224698fcu;
};
template <> struct Mod<FQ> {  // Pallas scalar / Vesta base
    static constexpr u32 P1 = 0x8c46eb21u, P2 = 0x0994a8ddu, P3 = 0x224698fcu;
};
static constexpr u32 P7 = 0x40000000u;

template <int F> __device__ __forceinline__ u32 mod_limb(int i) {
    return i == 0 ? 1u : i == 1 ? Mod<F>::P1 : i == 2 ? Mod<F>::P2 : i == 3 ? Mod<F>::P3 : i == 7 ? P7 : 0u;
}

// R mod p and R^2 mod p (verified against the Python oracle)
template <int F> __device__ __forceinline__ fe fe_one() {
    if (F == FP) return fe{{0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}};
    return fe{{0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}};
}
template <int F> __device__ __forceinline__ fe fe_r2() {
    if (F == FP) return fe{{0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu}};
    return fe{{0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu}};
}
__device__ __forceinline__ fe fe_zero() { return fe{{0, 0, 0, 0, 0, 0, 0, 0}}; }

__device__ __forceinline__ bool fe_is_zero(const fe &a) {
    return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0;
}
__device__ __forceinline__ bool fe_eq(const fe &a, const fe &b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

// Carry chains are written with __builtin_addc / __builtin_subc: they lower to v_add_co_u32 / v_addc_co_u32
// (one instruction per limb).  Spelling them with 64-bit temporaries makes hipcc emit v_lshl_add_u64 + v_mov
// shuffles instead -- ~5 instructions per limb, which made add/sub cost half a multiplication.

// r = a - p if a >= p else a          (a < 2p)
template <int F> __device__ __forceinline__ fe fe_reduce_once(const fe &a) {
    fe d;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 bo;
        d.v[i] = __builtin_subc(a.v[i], mod_limb<F>(i), br, &bo);
        br = bo;
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
    return r;
}

template <int F> __device__ __forceinline__ fe fe_add(const fe &a, const fe &b) {
    fe s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 co;
        s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
        c = co;
    }
    return fe_reduce_once<F>(s);  // a + b < 2p < 2^256: no carry out of limb 7
}

template <int F> __device__ __forceinline__ fe fe_sub(const fe &a, const fe &b) {
    fe d;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 bo;
        d.v[i] = __builtin_subc(a.v[i], b.v[i], br, &bo);
        br = bo;
    }
    // add p back when the subtraction borrowed
    const u32 mask = 0u - br;
    fe r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 co;
        r.v[i] = __builtin_addc(d.v[i], mod_limb<F>(i) & mask, c, &co);
        c = co;
    }
    return r;
}

template <int F> __device__ __forceinline__ fe fe_neg(const fe &a) { return fe_sub<F>(fe_zero(), a); }
template <int F> __device__ __forceinline__ fe fe_dbl(const fe &a) { return fe_add<F>(a, a); }

// ---- Montgomery multiplication ------------------------------------------------------------
#ifdef H2_FIELD_EXPERIMENTS
// Earlier multiplier variants, compiled only into tests/native/field_check.hip for A/B measurements.  They issue
// v_mad_u64_u32 -> v_addc_co_u32 back to back inside asm text, i.e. they lean on hardware interlocks for the
// gfx940/gfx950 "VALU writes SGPR -> VALU reads it" hazard; the shipped multiplier (fe_mul_sched) does not.
// 96-bit column accumulator step: {acc, hi} += x * y
__device__ __forceinline__ void mac96(u64 &acc, u32 &hi, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(hi)
        : "v"(x), "v"(y)
        : "vcc");
}
// same with a compile-time constant multiplier (modulus limb): constant goes through an SGPR/literal
__device__ __forceinline__ void mac96k(u64 &acc, u32 &hi, u32 x, u32 k) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(hi)
        : "v"(x), "s"(k)
        : "vcc");
}

template <int F> __device__ __forceinline__ fe fe_mul(const fe &a, const fe &b) {
    u32 m[8];
    fe r;
    u64 acc = 0;
    u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        // partial products of column k
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j >= 0 && j < 8) mac96(acc, hi, a.v[i], b.v[j]);
        }
        // reduction products m[i] * p[k-i], p = [1, P1, P2, P3, 0, 0, 0, 2^30]
        if (k >= 1 && k - 1 < 8) mac96k(acc, hi, m[k - 1], Mod<F>::P1);
        if (k >= 2 && k - 2 < 8) mac96k(acc, hi, m[k - 2], Mod<F>::P2);
        if (k >= 3 && k - 3 < 8) mac96k(acc, hi, m[k - 3], Mod<F>::P3);
        if (k >= 7 && k - 7 < 8) mac96k(acc, hi, m[k - 7], P7);
        u32 lo = (u32)acc, mid = (u32)(acc >> 32);
        if (k < 8) {
            // quotient digit m = -lo (since -p^-1 = -1 mod 2^32); adding m * p[0] = m clears lo and
            // carries 1 into mid exactly when lo != 0
            m[k] = 0u - lo;
            u32 c = lo != 0;
            u64 t = (u64)mid + c;
            acc = ((u64)(hi + (u32)(t >> 32)) << 32) | (u32)t;
        } else {
            r.v[k - 8] = lo;
            acc = ((u64)hi << 32) | mid;
        }
        hi = 0;
    }
    return fe_reduce_once<F>(r);  // (ab + mp)/R < 2p
}

// Variant: one asm statement per column (generated, see gen_field_mul.py)
template <int F> __device__ __forceinline__ fe fe_mul_col(const fe &a, const fe &b) {
#include "field_mul.inc"
    return fe_reduce_once<F>(r);
}

#ifdef H2_FIELD_EXPERIMENTS
// Variant: the whole multiplication as ONE asm statement with the accumulator in pinned VGPRs
// (generated, see gen_field_mul.py block_multiplier): no compiler glue between columns.
template <int F> __device__ __forceinline__ fe fe_mul_blk(const fe &a, const fe &b) {
#include "field_mul_blk.inc"
    return fe_reduce_once<F>(r);
}

#endif  // H2_FIELD_EXPERIMENTS

#endif  // H2_FIELD_EXPERIMENTS

// Variant: ONE asm statement, list-scheduled by gen_field_mul.py so that every carry (an SGPR written by a
// VALU instruction) is read no sooner than three instructions later -- the gfx940/gfx950 "VALU writes SGPR ->
// VALU reads it" hazard (2 wait states), which hipcc pads in its own code but cannot see inside asm text.
template <int F> __device__ __forceinline__ fe fe_mul_sched(const fe &a, const fe &b) {
#include "field_mul_sched.inc"
    return fe_reduce_once<F>(r);
}

// ---- lazy reduction for long chains (the MSM bucket accumulation) ---------------------------------------------------
// The conditional subtraction after every multiplication is 8 borrow-chained subtracts + 8 selects + the wait states the
// carries need: 28 issue slots of 276.  Dropping it leaves products in [0, 2p + d): with a, b < 2p + d the Montgomery
// quotient gives (ab + mp) / R < p + ab / R, and ab / R <= p (1 + e) + (d_a + d_b) / 2 with e = (4p - R) / R ~ 2^-128
// (both Pasta primes sit just ABOVE 2^254, so R = 2^256 misses the classical "R > 4p, no final subtraction" condition by
// a hair): the excess over 2p grows by at most p e ~ 2^126 per level of dependent multiplications, i.e. stays far below
// the 2^254 of headroom to 2^256 for any chain a lane can execute.  Differences use 2p as the bias.
template <int F> __device__ __forceinline__ u32 mod2_limb(int i) {   // 2p
    if (F == FP) { const u32 v[8] = {0x00000002u, 0x325a61dau, 0x1299f237u, 0x448d31f8u, 0, 0, 0, 0x80000000u}; return v[i]; }
    const u32 v[8] = {0x00000002u, 0x188dd642u, 0x132951bbu, 0x448d31f8u, 0, 0, 0, 0x80000000u};
    return v[i];
}
template <int F> __device__ __forceinline__ fe fe_mul_lazy(const fe &a, const fe &b) {
#include "field_mul_sched.inc"
    return r;
}
// a - b for lazy operands: a - b + 2p when the difference is negative.  That is non-negative whenever b <= 2p; a lazy
// value only reaches [2^255, 2p + d) when its canonical residue lies within ~2^128 of p (probability ~2^-126), and such a
// subtrahend is made canonical first -- a uniform, practically never taken loop, kept as a loop so that the compiler does
// not if-convert it into a third carry chain.
template <int F> __device__ __forceinline__ fe fe_sub_lazy(const fe &a, const fe &b_in) {
    fe b = b_in;
    while (b.v[7] >> 31) b = fe_reduce_once<F>(b);
    fe d;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 bo;
        d.v[i] = __builtin_subc(a.v[i], b.v[i], br, &bo);
        br = bo;
    }
    const u32 mask = 0u - br;
    fe r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 co;
        r.v[i] = __builtin_addc(d.v[i], mod2_limb<F>(i) & mask, c, &co);
        c = co;
    }
    return r;
}
// a + b for lazy operands: the sum may pass 2^256 (4p > 2^256), so the carry out of the top limb counts as ">= 2p"
template <int F> __device__ __forceinline__ fe fe_add_lazy(const fe &a, const fe &b) {
    fe s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 co;
        s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
        c = co;
    }
    fe d;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 bo;
        d.v[i] = __builtin_subc(s.v[i], mod2_limb<F>(i), br, &bo);
        br = bo;
    }
    const bool take = c != 0 || br == 0;      // true sum >= 2p
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = take ? d.v[i] : s.v[i];
    return r;
}
// canonical representative of a lazy value (< 3p)
template <int F> __device__ __forceinline__ fe fe_reduce_lazy(const fe &a) { return fe_reduce_once<F>(fe_reduce_once<F>(a)); }
// lazy value = 0 mod p ?
template <int F> __device__ __forceinline__ bool fe_is_zero_lazy(const fe &a) {
    u32 z = 0, e1 = 0, e2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        z |= a.v[i];
        e1 |= a.v[i] ^ mod_limb<F>(i);
        e2 |= a.v[i] ^ mod2_limb<F>(i);
    }
    return z == 0 || e1 == 0 || e2 == 0;
}

// Variant: plain C operand-scanning (CIOS); the compiler picks the instructions.  Kept as the
// readable specification of the multiplier and as an A/B baseline for the asm variants.
template <int F> __device__ __forceinline__ fe fe_mul_c(const fe &a, const fe &b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (u64)a.v[j] * b.v[i] + t[j] + (c >> 32);
            t[j] = (u32)c;
        }
        t[8] = (u32)(c >> 32);
        u32 m = 0u - t[0];
        u64 k = (t[0] != 0);
        c = (u64)m * Mod<F>::P1 + t[1] + k; t[0] = (u32)c;
        c = (u64)m * Mod<F>::P2 + t[2] + (c >> 32); t[1] = (u32)c;
        c = (u64)m * Mod<F>::P3 + t[3] + (c >> 32); t[2] = (u32)c;
        c = (u64)t[4] + (c >> 32); t[3] = (u32)c;
        c = (u64)t[5] + (c >> 32); t[4] = (u32)c;
        c = (u64)t[6] + (c >> 32); t[5] = (u32)c;
        c = ((u64)m << 30) + t[7] + (c >> 32); t[6] = (u32)c;
        c = (u64)t[8] + (c >> 32); t[7] = (u32)c;
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return fe_reduce_once<F>(r);
}

#ifndef H2_MUL_IMPL
#define H2_MUL_IMPL 4
#endif
template <int F> __device__ __forceinline__ fe fe_mulx(const fe &a, const fe &b) {
#if H2_MUL_IMPL == 0
    return fe_mul_c<F>(a, b);
#else
    return fe_mul_sched<F>(a, b);
#endif
}

template <int F> __device__ __forceinline__ fe fe_sqr(const fe &a) { return fe_mulx<F>(a, a); }

template <int F> __device__ __forceinline__ fe fe_to_mont(const fe &a) { return fe_mulx<F>(a, fe_r2<F>()); }
template <int F> __device__ __forceinline__ fe fe_from_mont(const fe &a) {
    return fe_mulx<F>(a, fe{{1, 0, 0, 0, 0, 0, 0, 0}});
}
// The same value by the reduction half alone (REDC of a 256-bit number): eight steps t <- (t + m p) / 2^32 with m = -t_0
// (-p^-1 = -1 mod 2^32) and p = [1, P1, P2, P3, 0, 0, 0, 2^30]: three multiply-adds and a carry ripple per step, ~120
// instructions and a dozen registers against the 248-instruction multiplier.  For the sort kernels, which turn every scalar
// into its canonical form before cutting digits and should stay small enough to share a SIMD with msm_accumulate.
template <int F> __device__ __forceinline__ fe fe_redc(const fe &a) {
    u32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
#pragma unroll
    for (int step = 0; step < 8; step++) {
        const u32 m = 0u - t[0];
        u64 acc = (u64)(t[0] != 0);                                            // t_0 + m = 0 or 2^32
        acc += (u64)t[1] + (u64)m * Mod<F>::P1;
        t[0] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[2] + (u64)m * Mod<F>::P2;
        t[1] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[3] + (u64)m * Mod<F>::P3;
        t[2] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[4];
        t[3] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[5];
        t[4] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[6];
        t[5] = (u32)acc;
        acc >>= 32;
        acc += (u64)t[7] + ((u64)m << 30);                                     // m * 2^30
        t[6] = (u32)acc;
        t[7] = (u32)(acc >> 32);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return fe_reduce_once<F>(r);                                               // (a + (R - 1) p) / R < p + 1
}

// a^(p-2) by square-and-multiply: ~256 squarings + ~80 products on the calling lane's dependent chain (~84 000 instructions).  Kept as the
// A/B arm (-DH2_FE_INV_FERMAT=1) of fe_inv below.
template <int F> __device__ fe fe_inv_fermat(const fe &a) {
    // exponent p - 2 = [0xffffffff, P1 - 1, P2, P3, 0, 0, 0, 2^30] (p0 = 1 borrows from P1)
    const u32 e[8] = {0xffffffffu, Mod<F>::P1 - 1, Mod<F>::P2, Mod<F>::P3, 0, 0, 0, P7};
    fe acc = fe_one<F>();
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr<F>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mulx<F>(acc, a);
    }
    return acc;
}
// 1 / a (Montgomery in, Montgomery out; 0 for 0): the divstep inversion of field_inv.cuh on the Montgomery word (a R)^-1 = a^-1 R^-1, brought
// back to a^-1 R by two products with R^2.  ~16 000 instructions against the ladder's ~84 000, constant time, no divergence between lanes.
#ifndef H2_FE_INV_FERMAT
#define H2_FE_INV_FERMAT 0
#endif
template <int F> __device__ fe fe_inv(const fe &a) {
#if H2_FE_INV_FERMAT
    return fe_inv_fermat<F>(a);
#else
    u32 p[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = mod_limb<F>(i);
    modinv30(a.v, p, y);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = y[i];
    const fe r2 = fe_r2<F>();
    return fe_mulx<F>(fe_mulx<F>(r, r2), r2);
#endif
}

// ---- memory access: one field element = 32 B = two 16-B vectors ------------------------------
__device__ __forceinline__ fe fe_load(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    return fe{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void fe_store(void *p, const fe &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

}  // namespace h2
