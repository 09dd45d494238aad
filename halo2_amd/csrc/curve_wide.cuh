// Lane-parallel point arithmetic for the latency-bound tails of the MSM (window fold, segment reduce, Horner).
//
// Those kernels run a handful of waves, each lane walking a chain of dependent point operations, and a lone
// wave already issues at ~75 % of the VALU peak: the chain length is the time.  Here one LOGICAL lane is a
// quad of 4 hardware lanes that hold identical copies of the operands; the independent field products of a
// point operation are computed one per lane in the same instruction stream and exchanged with DPP quad_perm
// broadcasts (one VALU move per limb; ds_bpermute shuffles measured slower than the multiplies they saved).
// XYZZ addition: 4 multiply levels instead of 14 sequential multiplies; doubling: 3 instead of 9.  Spare lanes are free here (the chip is >90 % idle in these kernels); the throughput-bound
// msm_accumulate keeps the one-lane-per-chain form.
#pragma once
#include "curve.cuh"

namespace h2 {

static constexpr int kGroup = 4;

// value of quad lane SRC in every lane of the quad: v_mov_b32_dpp quad_perm:[SRC,SRC,SRC,SRC]
template <int SRC> __device__ __forceinline__ fe g_bcast(const fe &r) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)r.v[i], ctrl, 0xf, 0xf, false);
    return o;
}
// lane-indexed operand pick, spelt with masks: a chain of `l == k ? a_k : ...` selects makes hipcc build a
// lookup table in scratch memory and index it per lane
__device__ __forceinline__ fe g_sel(int l, const fe &a0, const fe &a1, const fe &a2, const fe &a3) {
    // three bit-field inserts per limb: (m & a) | (~m & b) is one v_bfi_b32
    const u32 m0 = l == 0 ? ~0u : 0u, m1 = l == 1 ? ~0u : 0u, m2 = l == 2 ? ~0u : 0u;
    fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 t2 = (a2.v[i] & m2) | (a3.v[i] & ~m2);
        const u32 t1 = (a1.v[i] & m1) | (t2 & ~m1);
        o.v[i] = (a0.v[i] & m0) | (t1 & ~m0);
    }
    return o;
}
// 2 * p; every lane of the group passes the same p and receives the same result (dbl-2008-s-1, a = 0)
template <int F> __device__ __forceinline__ xyzz<F> xyzz_dbl_wide(const xyzz<F> &p) {
    if (xyzz_is_identity(p)) return p;
    const int l = threadIdx.x & (kGroup - 1);
    fe u = fe_dbl<F>(p.y);
    fe r1 = fe_mulx<F>(g_sel(l, u, p.x, u, u), g_sel(l, u, p.x, u, u));                 // lane0 V = U^2, lane1 XX = X^2
    fe v = g_bcast<0>(r1), xx = g_bcast<1>(r1);
    fe m = fe_add<F>(fe_dbl<F>(xx), xx);
    fe r2 = fe_mulx<F>(g_sel(l, u, p.x, m, v), g_sel(l, v, v, m, p.zz));                // W = U V, S = X V, MM = M^2, ZZ3 = V ZZ
    fe w = g_bcast<0>(r2), s = g_bcast<1>(r2), mm = g_bcast<2>(r2);
    xyzz<F> r;
    r.zz = g_bcast<3>(r2);
    r.x = fe_sub<F>(fe_sub<F>(mm, s), s);
    fe r3 = fe_mulx<F>(g_sel(l, m, w, w, w), g_sel(l, fe_sub<F>(s, r.x), p.y, p.zzz, p.zzz));   // M (S - X3), W Y, W ZZZ
    r.y = fe_sub<F>(g_bcast<0>(r3), g_bcast<1>(r3));
    r.zzz = g_bcast<2>(r3);
    return r;
}

// acc += q (add-2008-s), complete; group-uniform operands and result
template <int F> __device__ __forceinline__ void xyzz_add_wide(xyzz<F> &acc, const xyzz<F> &q) {
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc = q;
        return;
    }
    const int l = threadIdx.x & (kGroup - 1);
    fe r1 = fe_mulx<F>(g_sel(l, acc.x, q.x, acc.y, q.y), g_sel(l, q.zz, acc.zz, q.zzz, acc.zzz));           // U1, U2, S1, S2
    fe u1 = g_bcast<0>(r1), u2 = g_bcast<1>(r1), s1 = g_bcast<2>(r1), s2 = g_bcast<3>(r1);
    fe p = fe_sub<F>(u2, u1), rr = fe_sub<F>(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(rr)) acc = xyzz_dbl_wide<F>(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    fe r2 = fe_mulx<F>(g_sel(l, p, rr, acc.zz, acc.zzz), g_sel(l, p, rr, q.zz, q.zzz));                     // PP, R^2, ZZ1 ZZ2, ZZZ1 ZZZ2
    fe pp = g_bcast<0>(r2), r_sq = g_bcast<1>(r2), za = g_bcast<2>(r2), zb = g_bcast<3>(r2);
    fe r3 = fe_mulx<F>(g_sel(l, p, u1, za, za), pp);                                     // PPP, Q, ZZ3
    fe ppp = g_bcast<0>(r3), qq = g_bcast<1>(r3);
    acc.zz = g_bcast<2>(r3);
    fe x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(r_sq, ppp), qq), qq);
    fe r4 = fe_mulx<F>(g_sel(l, rr, s1, zb, zb), g_sel(l, fe_sub<F>(qq, x3), ppp, ppp, ppp));   // R (Q - X3), S1 PPP, ZZZ3
    acc.x = x3;
    acc.y = fe_sub<F>(g_bcast<0>(r4), g_bcast<1>(r4));
    acc.zzz = g_bcast<2>(r4);
}

// k * p, k < 2^16, group-uniform
template <int F> __device__ __forceinline__ xyzz<F> xyzz_mul_small_wide(const xyzz<F> &p, u32 k) {
    xyzz<F> r = xyzz_identity<F>();
    if (!k) return r;
    for (int b = 31 - __clz(k); b >= 0; --b) {
        r = xyzz_dbl_wide<F>(r);
        if ((k >> b) & 1) xyzz_add_wide<F>(r, p);
    }
    return r;
}

}  // namespace h2
