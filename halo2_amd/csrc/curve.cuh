// Pallas / Vesta group arithmetic on gfx950, register-resident, on top of field.cuh.
//
// Replaces the L0 curve ops the reference's MSM calls (pasta_curves Ep/Eq `+=`, `double`, used at
// halo2_proofs/src/arithmetic.rs:40-43,52-55,89-90,163).  Both curves are y^2 = x^3 + 5 (a = 0).
//
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed addition costs 8M + 2S instead of Jacobian's 7M + 4S and needs no inversion; identity is
// ZZ = 0.  All additions are complete: P + P, P + (-P), P + O and O + O are handled (the reference's
// L0 addition is complete), by wave-divergent but rare branches.
//
// Affine points in memory: {x, y}, 64 B, Montgomery limbs, identity = all-zero (what a Rust
// `EpAffine`/`EqAffine` holds).  Because y^2 = x^3 + 5 has no point with y = 0 over these fields' odd
// order groups and (0, 0) is not on the curve, all-zero is unambiguous.
#pragma once
#include "field.cuh"

namespace h2 {

template <int F> struct affine {
    fe x, y;
};
template <int F> struct xyzz {
    fe x, y, zz, zzz;
};

template <int F> __device__ __forceinline__ bool aff_is_identity(const affine<F> &p) {
    return fe_is_zero(p.x) && fe_is_zero(p.y);
}
template <int F> __device__ __forceinline__ bool xyzz_is_identity(const xyzz<F> &p) { return fe_is_zero(p.zz); }
template <int F> __device__ __forceinline__ xyzz<F> xyzz_identity() {
    return xyzz<F>{fe_zero(), fe_zero(), fe_zero(), fe_zero()};
}

// 2 * (affine P) -> XYZZ   (mdbl-2008-s-1, a = 0)
template <int F> __device__ __forceinline__ xyzz<F> xyzz_dbl_affine(const affine<F> &p) {
    fe u = fe_dbl<F>(p.y);
    fe v = fe_sqr<F>(u);
    fe w = fe_mulx<F>(u, v);
    fe s = fe_mulx<F>(p.x, v);
    fe xx = fe_sqr<F>(p.x);
    fe m = fe_add<F>(fe_dbl<F>(xx), xx);
    xyzz<F> r;
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(m), s), s);
    r.y = fe_sub<F>(fe_mulx<F>(m, fe_sub<F>(s, r.x)), fe_mulx<F>(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * (XYZZ P)   (dbl-2008-s-1, a = 0)
template <int F> __device__ __forceinline__ xyzz<F> xyzz_dbl(const xyzz<F> &p) {
    if (xyzz_is_identity(p)) return p;
    fe u = fe_dbl<F>(p.y);
    fe v = fe_sqr<F>(u);
    fe w = fe_mulx<F>(u, v);
    fe s = fe_mulx<F>(p.x, v);
    fe xx = fe_sqr<F>(p.x);
    fe m = fe_add<F>(fe_dbl<F>(xx), xx);
    xyzz<F> r;
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(m), s), s);
    r.y = fe_sub<F>(fe_mulx<F>(m, fe_sub<F>(s, r.x)), fe_mulx<F>(w, p.y));
    r.zz = fe_mulx<F>(v, p.zz);
    r.zzz = fe_mulx<F>(w, p.zzz);
    return r;
}

// acc += affine q   (madd-2008-s); complete
template <int F> __device__ __forceinline__ void xyzz_madd(xyzz<F> &acc, const affine<F> &q) {
    if (aff_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = fe_one<F>();
        acc.zzz = fe_one<F>();
        return;
    }
    fe u2 = fe_mulx<F>(q.x, acc.zz);
    fe s2 = fe_mulx<F>(q.y, acc.zzz);
    fe p = fe_sub<F>(u2, acc.x);
    fe r = fe_sub<F>(s2, acc.y);
    if (fe_is_zero(p)) {  // same x: doubling or inverse pair (rare; duplicate bases)
        if (fe_is_zero(r)) acc = xyzz_dbl_affine<F>(q);
        else acc = xyzz_identity<F>();
        return;
    }
    fe pp = fe_sqr<F>(p);
    fe ppp = fe_mulx<F>(p, pp);
    fe qq = fe_mulx<F>(acc.x, pp);
    fe x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), ppp), qq), qq);
    fe y3 = fe_sub<F>(fe_mulx<F>(r, fe_sub<F>(qq, x3)), fe_mulx<F>(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mulx<F>(acc.zz, pp);
    acc.zzz = fe_mulx<F>(acc.zzz, ppp);
}

// acc += affine q with LAZY accumulator coordinates (field.cuh: products are not conditionally subtracted, values live in
// [0, 2p + d)); q is canonical.  The bucket accumulation runs tens of these back to back per lane; xyzz_reduce_lazy makes
// the accumulator canonical again before it leaves the lane.
template <int F> __device__ __forceinline__ void xyzz_madd_lazy(xyzz<F> &acc, const affine<F> &q) {
    if (aff_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {       // the identity is stored as exact zeros
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = fe_one<F>();
        acc.zzz = fe_one<F>();
        return;
    }
    fe u2 = fe_mul_lazy<F>(q.x, acc.zz);
    fe s2 = fe_mul_lazy<F>(q.y, acc.zzz);
    fe p = fe_sub_lazy<F>(u2, acc.x);
    fe r = fe_sub_lazy<F>(s2, acc.y);
    if (fe_is_zero_lazy<F>(p)) {  // same x: doubling or inverse pair (rare; duplicate bases)
        if (fe_is_zero_lazy<F>(r)) acc = xyzz_dbl_affine<F>(q);
        else acc = xyzz_identity<F>();
        return;
    }
    fe pp = fe_mul_lazy<F>(p, p);
    fe ppp = fe_mul_lazy<F>(p, pp);
    fe qq = fe_mul_lazy<F>(acc.x, pp);
    fe x3 = fe_sub_lazy<F>(fe_sub_lazy<F>(fe_sub_lazy<F>(fe_mul_lazy<F>(r, r), ppp), qq), qq);
    fe y3 = fe_sub_lazy<F>(fe_mul_lazy<F>(r, fe_sub_lazy<F>(qq, x3)), fe_mul_lazy<F>(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul_lazy<F>(acc.zz, pp);
    acc.zzz = fe_mul_lazy<F>(acc.zzz, ppp);
}
template <int F> __device__ __forceinline__ void xyzz_reduce_lazy(xyzz<F> &acc) {
    acc.x = fe_reduce_lazy<F>(acc.x);
    acc.y = fe_reduce_lazy<F>(acc.y);
    acc.zz = fe_reduce_lazy<F>(acc.zz);
    acc.zzz = fe_reduce_lazy<F>(acc.zzz);
}

// acc += XYZZ q   (add-2008-s); complete
template <int F> __device__ __forceinline__ void xyzz_add(xyzz<F> &acc, const xyzz<F> &q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc = q;
        return;
    }
    fe u1 = fe_mulx<F>(acc.x, q.zz);
    fe u2 = fe_mulx<F>(q.x, acc.zz);
    fe s1 = fe_mulx<F>(acc.y, q.zzz);
    fe s2 = fe_mulx<F>(q.y, acc.zzz);
    fe p = fe_sub<F>(u2, u1);
    fe r = fe_sub<F>(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_dbl<F>(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    fe pp = fe_sqr<F>(p);
    fe ppp = fe_mulx<F>(p, pp);
    fe qq = fe_mulx<F>(u1, pp);
    fe x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), ppp), qq), qq);
    fe y3 = fe_sub<F>(fe_mulx<F>(r, fe_sub<F>(qq, x3)), fe_mulx<F>(s1, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mulx<F>(fe_mulx<F>(acc.zz, q.zz), pp);
    acc.zzz = fe_mulx<F>(fe_mulx<F>(acc.zzz, q.zzz), ppp);
}

// XYZZ -> Jacobian (X', Y', Z') with Z' = ZZ*ZZZ: X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2.  No inversion.
// This is the `C::Curve` the reference's best_multiexp returns (any representative of the point).
template <int F> __device__ __forceinline__ void xyzz_to_jacobian(const xyzz<F> &p, fe &X, fe &Y, fe &Z) {
    if (xyzz_is_identity(p)) {
        X = fe_zero();
        Y = fe_zero();
        Z = fe_zero();
        return;
    }
    fe z = fe_mulx<F>(p.zz, p.zzz);
    fe t = fe_mulx<F>(z, p.zzz);          // ZZ*ZZZ^2
    X = fe_mulx<F>(p.x, t);
    Y = fe_mulx<F>(p.y, fe_mulx<F>(t, fe_sqr<F>(p.zz)));
    Z = z;
}

// XYZZ -> affine (one field inversion).  identity -> (0, 0)
template <int F> __device__ inline affine<F> xyzz_to_affine(const xyzz<F> &p) {
    affine<F> r;
    if (xyzz_is_identity(p)) {
        r.x = fe_zero();
        r.y = fe_zero();
        return r;
    }
    fe i = fe_inv<F>(fe_mulx<F>(p.zz, p.zzz));    // 1/(ZZ*ZZZ)
    r.x = fe_mulx<F>(p.x, fe_mulx<F>(i, p.zzz));   // X/ZZ
    r.y = fe_mulx<F>(p.y, fe_mulx<F>(i, p.zz));    // Y/ZZZ
    return r;
}

template <int F> __device__ __forceinline__ affine<F> aff_load(const void *p) {
    const u32 *q = reinterpret_cast<const u32 *>(p);
    affine<F> r;
    r.x = fe_load(q);
    r.y = fe_load(q + 8);
    return r;
}
template <int F> __device__ __forceinline__ xyzz<F> xyzz_load(const void *p) {
    const u32 *q = reinterpret_cast<const u32 *>(p);
    xyzz<F> r;
    r.x = fe_load(q);
    r.y = fe_load(q + 8);
    r.zz = fe_load(q + 16);
    r.zzz = fe_load(q + 24);
    return r;
}
template <int F> __device__ __forceinline__ void xyzz_store(void *p, const xyzz<F> &a) {
    u32 *q = reinterpret_cast<u32 *>(p);
    fe_store(q, a.x);
    fe_store(q + 8, a.y);
    fe_store(q + 16, a.zz);
    fe_store(q + 24, a.zzz);
}

}  // namespace h2
