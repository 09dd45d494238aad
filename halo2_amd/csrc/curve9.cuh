// XYZZ mixed addition on the carry-free 9 x 29-bit field layer (field9.cuh): the inner loop of the MSM bucket accumulation.
// Same formulas and the same completeness as curve.cuh's xyzz_madd (madd-2008-s: 8M + 2S; P + P, P - P, O handled by rare
// divergent branches); coordinates are in M9 form (x * 2^261 mod p), normalised limbs, never conditionally subtracted.
//
// Limb / value bounds through one addition (N = normalised: limbs 0..7 in [0, 2^29), |value| < 2^256):
//   u2, s2, pp, ppp, qq, zz', zzz'   products                       -> N
//   p = u2 - X1, r = s2 - Y1         N - N: |limbs| < 2^29, |value| < 2^257                  (squared / multiplied by N: fine)
//   x3 = r^2 - ppp - 2 qq            fe9_sqr_minus: limbs normalised, |value| < 2^258            (it is multiplied by r below)
//   y3 = r (qq - x3) - Y1 ppp        fe9_dot2 (one reduction for both products): |r (qq - x3)| < 2^257 * 1.25 * 2^258 and
//                                    |Y1 ppp| < 2^512, sum < 2^516 -> N                         (it meets s2 in the next r)
// so no carry pass at all: the two results that used to need one leave the multiplier normalised.
#pragma once
#include "curve.cuh"
#include "field9.cuh"

namespace h2 {

template <int F> struct aff9 {
    fe9 x, y;
};
template <int F> struct xyzz9 {
    fe9 x, y, zz, zzz;
};

__device__ __forceinline__ bool fe9_all_zero(const fe9 &a) {
    i32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i];
    return o == 0;
}
template <int F> __device__ __forceinline__ xyzz9<F> xyzz9_identity() { return xyzz9<F>{fe9_zero(), fe9_zero(), fe9_zero(), fe9_zero()}; }
template <int F> __device__ __forceinline__ bool xyzz9_is_identity(const xyzz9<F> &p) { return fe9_all_zero(p.zz); }
// Identities are always stored as exact zero limbs (never as the result of a multiplication: a product of value 0 may come out
// as [2^29, 2^29 - 1, ..., -1]); aff9_unpack / aff9_from_r256 / xyzz9_from_r256 preserve that.
template <int F> __device__ __forceinline__ bool aff9_is_identity(const aff9<F> &p) { return fe9_all_zero(p.x) && fe9_all_zero(p.y); }

// a table entry stored in M9 form (canonical value packed 8 x 32): repacking only
template <int F> __device__ __forceinline__ aff9<F> aff9_unpack(const affine<F> &p) { return aff9<F>{fe9_unpack(p.x), fe9_unpack(p.y)}; }
// a point in the reference's Montgomery form (R = 2^256): one multiplication per coordinate.  (0, 0) stays (0, 0).
template <int F> __device__ __forceinline__ aff9<F> aff9_from_r256(const affine<F> &p) {
    if (aff_is_identity(p)) return aff9<F>{fe9_zero(), fe9_zero()};      // a PRODUCT with value 0 need not have all-zero limbs
    return aff9<F>{fe9_from_r256<F>(p.x), fe9_from_r256<F>(p.y)};
}
template <int F> __device__ __forceinline__ xyzz<F> xyzz9_to_r256(const xyzz9<F> &p) {
    return xyzz<F>{fe9_to_r256<F>(p.x), fe9_to_r256<F>(p.y), fe9_to_r256<F>(p.zz), fe9_to_r256<F>(p.zzz)};
}
template <int F> __device__ __forceinline__ xyzz9<F> xyzz9_from_r256(const xyzz<F> &p) {
    if (xyzz_is_identity(p)) return xyzz9_identity<F>();
    return xyzz9<F>{fe9_from_r256<F>(p.x), fe9_from_r256<F>(p.y), fe9_from_r256<F>(p.zz), fe9_from_r256<F>(p.zzz)};
}

// reference Montgomery form (x * 2^256) -> M9 form (x * 2^261), canonical, in the 8 x 32 layer: one multiplication by 2^5.
// Used where tables are built (registration, blind base); (0, 0) stays (0, 0).
template <int F> __device__ __forceinline__ affine<F> aff_to_m9(const affine<F> &p) {
    return affine<F>{fe_mulx<F>(p.x, fe_k32<F>()), fe_mulx<F>(p.y, fe_k32<F>())};
}

// raw limbs of an accumulator (36 words, 144 B): what msm_accumulate parks for msm_segments_to_r256
template <int F> __device__ __forceinline__ void xyzz9_store_raw(u32 *dst, const xyzz9<F> &a) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
    const fe9 *c[4] = {&a.x, &a.y, &a.zz, &a.zzz};
    u32 w[36];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 9; i++) w[9 * k + i] = (u32)c[k]->v[i];
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <int F> __device__ __forceinline__ xyzz9<F> xyzz9_load_raw(const u32 *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    xyzz9<F> a;
    fe9 *c[4] = {&a.x, &a.y, &a.zz, &a.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 9; i++) c[k]->v[i] = (i32)w[9 * k + i];
    return a;
}

// rare: the cheap filter on p = u2 - X1 fired.  Decides whether p really is 0 mod p and, if so, produces the sum (2q or the
// identity).  Everything goes in and out BY VALUE: handing the accumulator itself to a non-inlined function by reference would
// make it escape, and the compiler would then keep it in scratch memory through the whole accumulation loop.
template <int F> __device__ __noinline__ bool xyzz9_madd_rare(fe9 p, fe9 r, aff9<F> q, xyzz9<F> *out) {
    if (!fe_is_zero(fe9_canonical<F>(p))) return false;
    if (!fe_is_zero(fe9_canonical<F>(r))) {
        *out = xyzz9_identity<F>();
        return true;
    }
    affine<F> q256;
    q256.x = fe9_to_r256<F>(q.x);
    q256.y = fe9_to_r256<F>(q.y);
    *out = xyzz9_from_r256<F>(xyzz_dbl_affine<F>(q256));
    return true;
}

// acc += q; complete.  acc normalised on entry and exit.
// HOT = the caller's accumulation loop (msm_accumulate): q is known not to be the identity (the caller tested the packed point
// before unpacking it), and the accumulator is the identity exactly when limb 0 of its ZZ is zero -- every ZZ such a loop ever
// holds is all-zero (the identity), the constant one (limb 0 = 0x1fffff81) or a PRODUCT, whose limb 0 lies in [1, 2^29]
// (field9.cuh: the pending + 1 of the multiplier lands there).  Saves the second identity test on 18 limbs and seven of the nine
// ors of the first, per addition.
template <int F, bool HOT = false> __device__ __forceinline__ void xyzz9_madd(xyzz9<F> &acc, const aff9<F> &q) {
    if (!HOT && aff9_is_identity(q)) return;
    // (the one-limb test leans on the GENERATED multiplier's limb 0 in [1, 2^29]; the plain-C multiplier of the -DH2_FE9_IMPL=0 A/B
    // build returns limb 0 = acc & M29, zero once in 2^29 products: that build takes the full test)
    if ((HOT && H2_FE9_IMPL != 0) ? acc.zz.v[0] == 0 : xyzz9_is_identity(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = fe9_one_here<F>();      // (as plain constants hipcc hoists their 12 v_mov to the top of the accumulation loop: every
        acc.zzz = acc.zz;                //  iteration paid for what the first entry of a bucket needs)
        return;
    }
    const fe9 u2 = fe9_mul<F>(q.x, acc.zz);
    const fe9 s2 = fe9_mul<F>(q.y, acc.zzz);
    const fe9 p = fe9_sub(u2, acc.x);
    const fe9 r = fe9_sub(s2, acc.y);
    if (fe9_maybe_zero_mod_p(p)) {              // 3 instructions; true for 33 of 2^29 residues
        xyzz9<F> special;
        if (xyzz9_madd_rare<F>(p, r, q, &special)) {
            acc = special;
            return;
        }
    }
    const fe9 pp = fe9_sqr<F>(p);
    const fe9 ppp = fe9_mul<F>(p, pp);
    const fe9 qq = fe9_mul<F>(acc.x, pp);
    // X3 = R^2 - PPP - 2 Q and Y3 = R (Q - X3) - Y1 PPP in the fused forms (field9.cuh): the subtrahend of X3 rides in the
    // square's upper columns, the two products of Y3 share one reduction -- 138 instructions fewer than two products, a square,
    // three subtractions and two carry passes, and both results leave normalised
    const fe9 x3 = fe9_sqr_minus<F>(r, fe9_add(fe9_dbl(qq), ppp));
    const fe9 y3 = fe9_dot2<F>(r, fe9_sub(qq, x3), fe9_sub(fe9_zero(), acc.y), ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe9_mul<F>(acc.zz, pp);
    acc.zzz = fe9_mul<F>(acc.zzz, ppp);
}

// ---- full XYZZ + XYZZ addition on the same layer (add-2008-s: 12M + 2S): the throughput form of the bucket fold --------------
// rare: p = U2 - U1 passed the cheap filter.  By value, as xyzz9_madd_rare (see there).
template <int F> __device__ __noinline__ bool xyzz9_add_rare(fe9 p, fe9 r, xyzz9<F> a, xyzz9<F> *out) {
    if (!fe_is_zero(fe9_canonical<F>(p))) return false;
    if (!fe_is_zero(fe9_canonical<F>(r))) {
        *out = xyzz9_identity<F>();
        return true;
    }
    *out = xyzz9_from_r256<F>(xyzz_dbl<F>(xyzz9_to_r256<F>(a)));      // the two operands are the same point
    return true;
}
// acc += q; complete.  Both normalised on entry, acc normalised on exit (same bounds discipline as xyzz9_madd).
template <int F> __device__ __forceinline__ void xyzz9_add(xyzz9<F> &acc, const xyzz9<F> &q) {
    if (xyzz9_is_identity(q)) return;
    if (xyzz9_is_identity(acc)) {
        acc = q;
        return;
    }
    const fe9 u1 = fe9_mul<F>(acc.x, q.zz), u2 = fe9_mul<F>(q.x, acc.zz);
    const fe9 s1 = fe9_mul<F>(acc.y, q.zzz), s2 = fe9_mul<F>(q.y, acc.zzz);
    const fe9 p = fe9_sub(u2, u1), r = fe9_sub(s2, s1);
    if (fe9_maybe_zero_mod_p(p)) {
        xyzz9<F> special;
        // the OTHER operand goes to the out-of-line function (when it matters the two are the same point): handing over the
        // loop-carried accumulator made hipcc keep it in scratch memory through every caller's loop (36 words stored and
        // reloaded per addition -- 78 MB of scratch writes per launch of the line-sum kernel)
        if (xyzz9_add_rare<F>(p, r, q, &special)) {
            acc = special;
            return;
        }
    }
    const fe9 pp = fe9_sqr<F>(p);
    const fe9 ppp = fe9_mul<F>(p, pp);
    const fe9 qq = fe9_mul<F>(u1, pp);
    const fe9 x3 = fe9_sqr_minus<F>(r, fe9_add(fe9_dbl(qq), ppp));
    const fe9 y3 = fe9_dot2<F>(r, fe9_sub(qq, x3), fe9_sub(fe9_zero(), s1), ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe9_mul<F>(fe9_mul<F>(acc.zz, q.zz), pp);
    acc.zzz = fe9_mul<F>(fe9_mul<F>(acc.zzz, q.zzz), ppp);
}

}  // namespace h2
