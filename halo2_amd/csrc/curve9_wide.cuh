// Quad-lane point arithmetic on the carry-free 9 x 29-bit layer: the fe9 counterpart of curve_wide.cuh.
//
// curve_wide.cuh cuts the latency of a dependent point operation by spreading its independent field products over a quad of
// lanes; each product there is the 8 x 32 multiplier (248 instructions + a conditional subtraction) and the operand selects
// and broadcasts cost 8 words each.  The same scheme on the carry-free layer has a 165-instruction product and no reduction
// step: a doubling is ~850 instructions against ~1300.  That matters exactly where a long chain of doublings is the whole
// cost: the Horner step over the window slices of a generic multiexp (msm_combine: 128-130 doublings, more than half the time of
// every multiexp below 2^16 points) and the doubling chains of small registered tables.
// Every lane of a quad passes the same operands and receives the same result; values are in M9 form, normalised
// (curve9.cuh's bounds discipline: products are normalised, sums / differences of a few products are valid operands, values
// that feed another product after more than one subtraction take a carry pass).
#pragma once
#include "curve9.cuh"
#include "curve_wide.cuh"

namespace h2 {

template <int SRC> __device__ __forceinline__ fe9 g9_bcast(const fe9 &r) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    fe9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.v[i] = __builtin_amdgcn_mov_dpp(r.v[i], ctrl, 0xf, 0xf, false);
    return o;
}
// lane-indexed operand pick with masks (see g_sel in curve_wide.cuh)
__device__ __forceinline__ fe9 g9_sel(int l, const fe9 &a0, const fe9 &a1, const fe9 &a2, const fe9 &a3) {
    const u32 m0 = l == 0 ? ~0u : 0u, m1 = l == 1 ? ~0u : 0u, m2 = l == 2 ? ~0u : 0u;
    fe9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 t2 = ((u32)a2.v[i] & m2) | ((u32)a3.v[i] & ~m2);
        const u32 t1 = ((u32)a1.v[i] & m1) | (t2 & ~m1);
        o.v[i] = (i32)(((u32)a0.v[i] & m0) | (t1 & ~m0));
    }
    return o;
}

// 2 p (dbl-2008-s-1, a = 0): three product levels
template <int F> __device__ __forceinline__ xyzz9<F> xyzz9_dbl_wide(const xyzz9<F> &p) {
    if (xyzz9_is_identity(p)) return p;
    const int l = threadIdx.x & (kGroup - 1);
    const fe9 u = fe9_dbl(p.y);                                                     // limbs < 2^30: fine against a normalised operand,
    const fe9 o1 = g9_sel(l, p.y, p.x, p.y, p.y);                                   // not against itself: V = U^2 is formed as 4 Y^2
    const fe9 r1 = fe9_mul<F>(o1, o1);                                              // lane 0: YY = Y^2, lane 1: XX = X^2
    const fe9 yy = g9_bcast<0>(r1), xx = g9_bcast<1>(r1);
    const fe9 v = fe9_quadruple_norm(yy);                                           // (a product's limb 0 may be 2^29: see there)
    const fe9 m = fe9_norm(fe9_add(fe9_dbl(xx), xx));                               // 3 XX
    const fe9 r2 = fe9_mul<F>(g9_sel(l, u, p.x, m, v), g9_sel(l, v, v, m, p.zz));   // W = U V, S = X V, MM = M^2, ZZ3 = V ZZ
    const fe9 w = g9_bcast<0>(r2), s = g9_bcast<1>(r2), mm = g9_bcast<2>(r2);
    xyzz9<F> r;
    r.zz = g9_bcast<3>(r2);
    r.x = fe9_norm(fe9_sub(fe9_sub(mm, s), s));
    const fe9 r3 = fe9_mul<F>(g9_sel(l, m, w, w, w), g9_sel(l, fe9_sub(s, r.x), p.y, p.zzz, p.zzz));   // M (S - X3), W Y, W ZZZ
    r.y = fe9_norm(fe9_sub(g9_bcast<0>(r3), g9_bcast<1>(r3)));
    r.zzz = g9_bcast<2>(r3);
    return r;
}

// acc += q (add-2008-s), complete: four product levels
template <int F> __device__ __forceinline__ void xyzz9_add_wide(xyzz9<F> &acc, const xyzz9<F> &q) {
    if (xyzz9_is_identity(q)) return;
    if (xyzz9_is_identity(acc)) {
        acc = q;
        return;
    }
    const int l = threadIdx.x & (kGroup - 1);
    const fe9 r1 = fe9_mul<F>(g9_sel(l, acc.x, q.x, acc.y, q.y), g9_sel(l, q.zz, acc.zz, q.zzz, acc.zzz));   // U1, U2, S1, S2
    const fe9 u1 = g9_bcast<0>(r1), u2 = g9_bcast<1>(r1), s1 = g9_bcast<2>(r1), s2 = g9_bcast<3>(r1);
    const fe9 p = fe9_sub(u2, u1), rr = fe9_sub(s2, s1);
    if (fe9_maybe_zero_mod_p(p)) {                       // rare; every lane of the quad takes the same one-lane path
        xyzz9<F> special;
        if (xyzz9_add_rare<F>(p, rr, q, &special)) {          // q, not acc: see xyzz9_add
            acc = special;
            return;
        }
    }
    const fe9 r2 = fe9_mul<F>(g9_sel(l, p, rr, acc.zz, acc.zzz), g9_sel(l, p, rr, q.zz, q.zzz));            // PP, R^2, ZZ1 ZZ2, ZZZ1 ZZZ2
    const fe9 pp = g9_bcast<0>(r2), r_sq = g9_bcast<1>(r2), za = g9_bcast<2>(r2), zb = g9_bcast<3>(r2);
    const fe9 r3 = fe9_mul<F>(g9_sel(l, p, u1, za, za), pp);                                                  // PPP, Q, ZZ3
    const fe9 ppp = g9_bcast<0>(r3), qq = g9_bcast<1>(r3);
    acc.zz = g9_bcast<2>(r3);
    const fe9 x3 = fe9_norm(fe9_sub(fe9_sub(r_sq, ppp), fe9_dbl(qq)));
    const fe9 r4 = fe9_mul<F>(g9_sel(l, rr, s1, zb, zb), g9_sel(l, fe9_sub(qq, x3), ppp, ppp, ppp));          // R (Q - X3), S1 PPP, ZZZ3
    acc.x = x3;
    acc.y = fe9_norm(fe9_sub(g9_bcast<0>(r4), g9_bcast<1>(r4)));
    acc.zzz = g9_bcast<2>(r4);
}

// reference Montgomery form <-> M9 form with the four coordinates on the four lanes: one product level each way
template <int F> __device__ __forceinline__ xyzz9<F> xyzz9_from_r256_wide(const xyzz<F> &p) {
    if (xyzz_is_identity(p)) return xyzz9_identity<F>();
    const int l = threadIdx.x & (kGroup - 1);
    const fe9 r = fe9_mul<F>(g9_sel(l, fe9_unpack(p.x), fe9_unpack(p.y), fe9_unpack(p.zz), fe9_unpack(p.zzz)), fe9_k_in<F>());
    return xyzz9<F>{g9_bcast<0>(r), g9_bcast<1>(r), g9_bcast<2>(r), g9_bcast<3>(r)};
}
template <int F> __device__ __forceinline__ xyzz<F> xyzz9_to_r256_wide(const xyzz9<F> &p) {
    if (xyzz9_is_identity(p)) return xyzz_identity<F>();
    const int l = threadIdx.x & (kGroup - 1);
    const fe9 r = fe9_mul<F>(g9_sel(l, p.x, p.y, p.zz, p.zzz), fe9_k_out<F>());
    const fe c = fe9_canonical_small<F>(r);                // each lane its own coordinate
    fe x, y, zz, zzz;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)c.v[i], 0x00, 0xf, 0xf, false);
        y.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)c.v[i], 0x55, 0xf, 0xf, false);
        zz.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)c.v[i], 0xAA, 0xf, 0xf, false);
        zzz.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)c.v[i], 0xFF, 0xf, 0xf, false);
    }
    return xyzz<F>{x, y, zz, zzz};
}

}  // namespace h2
