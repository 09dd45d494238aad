// Inner-product-argument round kernels: the per-round work of
// `commitment::create_proof` (halo2_proofs/src/poly/commitment/prover.rs:100-142) that is not an MSM.
//
//   h2_generator_collapse   parallel_generator_collapse (:154-166):  g'[i] = g_lo[i] + [u_j] * g_hi[i], normalised to
//                           affine -- n / 2^(j+1) FULL 255-bit scalar multiplications per round, an order of
//                           magnitude more group operations per proof than one commit.  The challenge is the same
//                           for every lane, so the host splits it once with the curve endomorphism
//                           phi(x, y) = (zeta x, y) = [lambda](x, y):  u = k1 + k2 lambda, |k1|, |k2| < 2^129, and
//                           the NAFs of k1 and k2 drive a divergence-free JOINT double-and-add over P and phi(P):
//                           ~130 doublings + ~87 mixed adds instead of 255 + 85.  One lane per point for large
//                           rounds; the last rounds (<= 2^15 points, pure latency) run one point per quad of lanes
//                           (curve_wide.cuh).  One Fermat inversion per point normalises.
//   h2_fold_scalars         the `p'` / `b` collapse (:128-131):  a[i] += a[i + half] * factor.
//
// Both keep their vectors on the device across rounds (d_* variants), removing 2k host round trips per proof.
#include <vector>

#include "common.h"
#include "curve_wide.cuh"
#include "glv.cuh"
#include "host_field.h"

namespace h2 {

int bases_refill_device(h2_bases_t handle, const void *d_bases_xy, size_t n, int form);      // msm.hip
int bases_register_device_internal(int curve, const void *d_bases_xy, size_t n, int form, h2_bases_t *handle, bool glv);
bool pair_subdigits_apply(size_t n);

// naf1 / naf2: signed digits in {-1, 0, 1} of k1 and k2 (signs folded in), little-endian, uniform across lanes;
// g[i] <- g[i] + [k1] g[half + i] + [k2] phi(g[half + i])
template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse(u32 *__restrict__ g, u32 half, const int8_t *__restrict__ naf1,
                                                    const int8_t *__restrict__ naf2, int top) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const affine<FB> hi = aff_load<FB>(g + 16 * (size_t)(half + i));
    const fe neg_y = fe_neg<FB>(hi.y);
    const fe phi_x = fe_mulx<FB>(hi.x, glv_zeta<FB>());
    xyzz<FB> acc = xyzz_identity<FB>();
    for (int b = top; b >= 0; --b) {           // uniform control flow: every lane walks the same digits
        acc = xyzz_dbl<FB>(acc);
        const int d1 = naf1[b], d2 = naf2[b];
        if (d1) xyzz_madd<FB>(acc, affine<FB>{hi.x, d1 > 0 ? hi.y : neg_y});
        if (d2) xyzz_madd<FB>(acc, affine<FB>{phi_x, d2 > 0 ? hi.y : neg_y});
    }
    const affine<FB> lo = aff_load<FB>(g + 16 * (size_t)i);
    xyzz_madd<FB>(acc, lo);
    const affine<FB> r = xyzz_to_affine<FB>(acc);
    fe_store(g + 16 * (size_t)i, r.x);
    fe_store(g + 16 * (size_t)i + 8, r.y);
}

// the same walk with one point per quad of lanes: the last rounds of an argument have a handful of points and are
// bound by the ~220 sequential point operations, which the quad runs at 3-4 multiplication levels each
template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse_wide(u32 *__restrict__ g, u32 half, const int8_t *__restrict__ naf1,
                                                         const int8_t *__restrict__ naf2, int top) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (i >= half) return;
    const affine<FB> hi = aff_load<FB>(g + 16 * (size_t)(half + i));
    const fe one = fe_one<FB>();
    const fe neg_y = fe_neg<FB>(hi.y);
    const fe phi_x = fe_mulx<FB>(hi.x, glv_zeta<FB>());
    xyzz<FB> acc = xyzz_identity<FB>();
    const bool hi_id = fe_is_zero(hi.x) && fe_is_zero(hi.y);      // the identity as an affine operand: nothing to add
    for (int b = top; b >= 0 && !hi_id; --b) {
        acc = xyzz_dbl_wide<FB>(acc);
        const int d1 = naf1[b], d2 = naf2[b];
        if (d1) xyzz_add_wide<FB>(acc, xyzz<FB>{hi.x, d1 > 0 ? hi.y : neg_y, one, one});
        if (d2) xyzz_add_wide<FB>(acc, xyzz<FB>{phi_x, d2 > 0 ? hi.y : neg_y, one, one});
    }
    const affine<FB> lo = aff_load<FB>(g + 16 * (size_t)i);
    if (!(fe_is_zero(lo.x) && fe_is_zero(lo.y))) xyzz_add_wide<FB>(acc, xyzz<FB>{lo.x, lo.y, one, one});
    if ((threadIdx.x & (kGroup - 1)) != 0) return;
    const affine<FB> r = xyzz_to_affine<FB>(acc);
    fe_store(g + 16 * (size_t)i, r.x);
    fe_store(g + 16 * (size_t)i + 8, r.y);
}

template <int F>
__global__ void __launch_bounds__(256) ipa_fold(u32 *__restrict__ a, u32 half, fe factor) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    fe lo = fe_load(a + 8 * (size_t)i), hi = fe_load(a + 8 * (size_t)(half + i));
    fe_store(a + 8 * (size_t)i, fe_add<F>(lo, fe_mulx<F>(hi, factor)));
}

template <int F> __global__ void __launch_bounds__(256) ipa_to_mont(u32 *a, size_t n, int to) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe v = fe_load(a + 8 * i);
    fe_store(a + 8 * i, to ? fe_to_mont<F>(v) : fe_from_mont<F>(v));
}

// ---- L_j / R_j over the ORIGINAL generators (h2_ipa_round_scalars_device) ----
// s_j(h) = prod_{r < j} u_r^{bit_{j-1-r}(h)}: round r's collapse pairs index bit k-1-r, which is bit j-1-r of h = m >> (k-j).
// (verifier.rs:156-172 compute_s builds the same products for the verifier.)  One lane per h, <= j multiplications.
template <int F>
__global__ void __launch_bounds__(256) ipa_s_table(u32 *__restrict__ s, const u32 *__restrict__ u_mont, u32 j) {
    const u32 h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >> j) return;
    fe acc = fe_one<F>();
    for (u32 r = 0; r < j; ++r)
        if ((h >> (j - 1 - r)) & 1) acc = fe_mulx<F>(acc, fe_load(u_mont + 8 * r));
    fe_store(s + 8 * (size_t)h, acc);
}

// cl[m] = p'[half + i] s_j(h) for i < half (else 0); cr[m] = p'[i - half] s_j(h) for i >= half (else 0); m = h 2^(k-j) + i.
// s is Montgomery, so the products keep whatever form p' is in.
template <int F>
__global__ void __launch_bounds__(256) ipa_round_scalars(const u32 *__restrict__ p, const u32 *__restrict__ s, u32 k, u32 j,
                                                         u32 *__restrict__ cl, u32 *__restrict__ cr) {
    const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >> k) return;
    const u32 blk = k - j, half = 1u << (blk - 1);
    const u32 h = m >> blk, i = m & ((1u << blk) - 1);
    const fe v = fe_mulx<F>(fe_load(p + 8 * (size_t)(i ^ half)), fe_load(s + 8 * (size_t)h));
    const bool lo = i < half;
    if (cl == cr) {        // merged column for a pair commit (h2_commit_pair_device): L_j and R_j have disjoint supports
        fe_store(cl + 8 * (size_t)m, v);
        return;
    }
    fe_store(cl + 8 * (size_t)m, lo ? v : fe_zero());
    fe_store(cr + 8 * (size_t)m, lo ? fe_zero() : v);
}

struct IpaContext {
    std::mutex mu;
    DevBuf naf, stage, stab;
    // the round loop (h2_ipa_rounds_device): its own lock (held for the whole argument; the launches it makes take `mu`),
    // device scratch { <p'_hi, b_lo>, <p'_lo, b_hi>, L_j, R_j } and the pinned landing pad of L_j, R_j
    std::mutex rounds_mu;
    DevBuf rounds, gprime, rstab;      // rstab: the round loop's two s tables (the single-round entry point keeps `stab`)
    void *rounds_host = nullptr;
    h2_bases_t gp_handle = 0;          // the table of the collapsed generators, kept between arguments (rebuilt in place while its shape repeats)
    size_t gp_n = 0;
    int gp_curve = -1;
    // the whole argument (h2_open_device / h2_open): b, the round loop's column(s), the landing places of host vectors, a few words
    // { s(x_3), p(x_3), the S commitment, its blind }, a pinned landing pad, and a side stream for what does not wait for xi
    std::mutex open_mu;
    DevBuf open_b, open_col, open_s, open_p, open_small;
    void *open_host = nullptr;
    hipStream_t open_side = nullptr, open_side2 = nullptr;
    hipEvent_t open_ev = nullptr, open_ev2 = nullptr;
    hipEvent_t open_land[4] = {nullptr, nullptr, nullptr, nullptr}, open_fixed = nullptr, open_parts = nullptr;     // the S commitment by ranges (open_impl)
    void release_all() {
        naf.release();
        stage.release();
        stab.release();
        // h2_trim must not take the round loop's scratch from under a running argument (which holds rounds_mu for its whole
        // length and takes `mu` inside: try, never wait -- the lock order is the other way round there)
        std::unique_lock<std::mutex> rl(rounds_mu, std::try_to_lock);
        if (!rl.owns_lock()) return;
        rounds.release();
        gprime.release();
        rstab.release();
        if (gp_handle) (void)h2_bases_free(gp_handle);
        gp_handle = 0;
        if (rounds_host) (void)hipHostFree(rounds_host);
        rounds_host = nullptr;
        std::unique_lock<std::mutex> ol(open_mu, std::try_to_lock);      // (as above: never under a running argument)
        if (!ol.owns_lock()) return;
        open_b.release();
        open_col.release();
        open_s.release();
        open_p.release();
        open_small.release();
        if (open_host) (void)hipHostFree(open_host);
        open_host = nullptr;
        if (open_ev) (void)hipEventDestroy(open_ev);
        if (open_ev2) (void)hipEventDestroy(open_ev2);
        open_ev = open_ev2 = nullptr;
        if (open_side) (void)hipStreamDestroy(open_side);
        if (open_side2) (void)hipStreamDestroy(open_side2);
        open_side = open_side2 = nullptr;
        for (hipEvent_t *e : {&open_land[0], &open_land[1], &open_land[2], &open_land[3], &open_fixed, &open_parts}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    }
};
static StreamContexts<IpaContext> g_ipa_ctxs;
void ipa_release_workspaces() { g_ipa_ctxs.release_current_device(); }   // h2_trim

// non-adjacent form of a canonical scalar below 2^256; returns the index of the top non-zero digit (-1 for zero)
static int naf_recode(const u64 k_in[4], int8_t out[257]) {
    u64 k[5] = {k_in[0], k_in[1], k_in[2], k_in[3], 0};
    memset(out, 0, 257);
    int top = -1;
    for (int i = 0; i < 257; ++i) {
        if (k[0] & 1) {
            int d = 2 - (int)(k[0] & 3);   // +1 if k = 1 mod 4, -1 if k = 3 mod 4
            out[i] = (int8_t)d;
            top = i;
            if (d > 0) {
                k[0] -= 1;
            } else {                        // k += 1 with carry
                for (int j = 0; j < 5; ++j)
                    if (++k[j] != 0) break;
            }
        }
        for (int j = 0; j < 4; ++j) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    return top;
}

// ---- GLV split of the challenge: u = k1 + k2 * lambda (mod the scalar-field modulus), |k1|, |k2| < 2^129 -----------
// Lattice basis (a1, b1), (a2, b2) with a + b * lambda = 0, and g_i = floor(2^256 * (b2, -b1) / q): all derived with
// big-integer arithmetic offline (extended Euclid on (q, lambda)); lambda is the root of X^2 + X + 1 with
// [lambda](x, y) = (zeta x, y) for the zeta in glv_zeta().  c_i = (u * g_i) >> 256 only has to be CLOSE to the exact
// quotient: any integers c1, c2 give k1 + k2 lambda = u; closeness keeps k1, k2 short.
struct GlvConst {
    u64 a1[2], b1_abs[2], a2[2], b2[2], g1[3], g2[3];   // b1 is negative for both curves, everything else positive
};
static const GlvConst kGlv[2] = {
    // scalar field Fq (Pallas)
    {{0x7fcae1c700000001ULL, 0x49e69d1640f04915ULL}, {0x8cb1279300000000ULL, 0x49e69d1640a89953ULL},
     {0x8cb1279300000000ULL, 0x49e69d1640a89953ULL}, {0x0c7c095a00000001ULL, 0x93cd3a2c8198e269ULL},
     {0x31f0256800000002ULL, 0x4f34e8b2066389a4ULL, 2}, {0x32c49e4bffffffffULL, 0x279a745902a2654eULL, 1}},
    // scalar field Fp (Vesta)
    {{0x8cb1279300000001ULL, 0x49e69d1640a89953ULL}, {0x7fcae1c700000000ULL, 0x49e69d1640f04915ULL},
     {0x0c7c095a00000001ULL, 0x93cd3a2c8198e269ULL}, {0x8cb1279300000001ULL, 0x49e69d1640a89953ULL},
     {0x32c49e4c00000003ULL, 0x279a745902a2654eULL, 1}, {0xff2b871bffffffffULL, 0x279a745903c12455ULL, 1}},
};
// out[na + nb] = a * b
static void limbs_mul(u64 *out, const u64 *a, int na, const u64 *b, int nb) {
    memset(out, 0, (size_t)(na + nb) * 8);
    for (int i = 0; i < na; ++i) {
        u128 carry = 0;
        for (int j = 0; j < nb; ++j) {
            carry += (u128)a[i] * b[j] + out[i + j];
            out[i + j] = (u64)carry;
            carry >>= 64;
        }
        out[i + nb] = (u64)carry;
    }
}
// 6-limb two's complement: r = a +/- b (b zero-extended from nb limbs)
static void acc6(u64 r[6], const u64 *b, int nb, bool subtract) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    memcpy(t, b, (size_t)nb * 8);
    unsigned carry = subtract ? 1 : 0;
    for (int i = 0; i < 6; ++i) {
        const u64 x = subtract ? ~t[i] : t[i];
        const u128 v = (u128)r[i] + x + carry;
        r[i] = (u64)v;
        carry = (unsigned)(v >> 64);
    }
}
// |v| of a 6-limb two's complement value into 4 limbs; returns true when v < 0
static bool abs6(const u64 v[6], u64 out[4]) {
    const bool neg = (v[5] >> 63) != 0;
    u64 t[6];
    memcpy(t, v, 48);
    if (neg) {
        unsigned carry = 1;
        for (int i = 0; i < 6; ++i) {
            const u128 w = (u128)(~t[i]) + carry;
            t[i] = (u64)w;
            carry = (unsigned)(w >> 64);
        }
    }
    memcpy(out, t, 32);   // |k_i| < 2^129
    return neg;
}
// digits of k1 and k2 with their signs folded in; returns the highest index used by either
static int glv_recode(int scalar_field, const u64 u_canonical[4], int8_t naf1[257], int8_t naf2[257]) {
    const GlvConst &G = kGlv[scalar_field == H2_FQ ? 0 : 1];
    u64 prod[7], c1[3], c2[3];
    limbs_mul(prod, u_canonical, 4, G.g1, 3);
    memcpy(c1, prod + 4, 24);
    limbs_mul(prod, u_canonical, 4, G.g2, 3);
    memcpy(c2, prod + 4, 24);
    // k1 = u - c1 a1 - c2 a2;   k2 = -c1 b1 - c2 b2 = c1 |b1| - c2 b2
    u64 k1[6] = {u_canonical[0], u_canonical[1], u_canonical[2], u_canonical[3], 0, 0}, k2[6] = {0, 0, 0, 0, 0, 0}, t[5];
    limbs_mul(t, c1, 3, G.a1, 2);
    acc6(k1, t, 5, true);
    limbs_mul(t, c2, 3, G.a2, 2);
    acc6(k1, t, 5, true);
    limbs_mul(t, c1, 3, G.b1_abs, 2);
    acc6(k2, t, 5, false);
    limbs_mul(t, c2, 3, G.b2, 2);
    acc6(k2, t, 5, true);
    u64 m1[4], m2[4];
    const bool n1 = abs6(k1, m1), n2 = abs6(k2, m2);
    int top1 = naf_recode(m1, naf1), top2 = naf_recode(m2, naf2);
    if (n1)
        for (int i = 0; i <= top1; ++i) naf1[i] = (int8_t)-naf1[i];
    if (n2)
        for (int i = 0; i <= top2; ++i) naf2[i] = (int8_t)-naf2[i];
    return top1 > top2 ? top1 : top2;
}

static int collapse_launch(int curve, void *d_g, size_t half, const u64 *u, int form, hipStream_t st) {
    IpaContext &cx = g_ipa_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const int sf = curve == H2_PALLAS ? H2_FQ : H2_FP;      // the challenge lives in the scalar field
    u64 canon[4];
    if (form == H2_FORM_MONTGOMERY) host_from_mont(sf, canon, u);
    else memcpy(canon, u, 32);
    int8_t naf[2 * 264];
    memset(naf, 0, sizeof naf);
    int top = glv_recode(sf, canon, naf, naf + 264);
    int rc = cx.naf.reserve(sizeof naf);
    if (rc != H2_OK) return rc;
    // the digit buffer belongs to this (device, stream): the copy is stream-ordered after the previous collapse that read
    // it, and hipMemcpyAsync from pageable memory has consumed `naf` when it returns
    H2_HIP(hipMemcpyAsync(cx.naf.ptr, naf, sizeof naf, hipMemcpyHostToDevice, st));
    const int8_t *d1 = cx.naf.as<int8_t>(), *d2 = d1 + 264;
    const bool wide = half <= 32768;      // few points: latency-bound, one point per quad of lanes
    dim3 grid((unsigned)(((wide ? half * kGroup : half) + 255) / 256)), block(256);
    if (form == H2_FORM_CANONICAL) {
        dim3 g2((unsigned)((half * 4 + 255) / 256));
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_to_mont<FP>), g2, block, 0, st, (u32 *)d_g, half * 4, 1);
        else hipLaunchKernelGGL((ipa_to_mont<FQ>), g2, block, 0, st, (u32 *)d_g, half * 4, 1);
    }
    if (wide) {
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_collapse_wide<FP>), grid, block, 0, st, (u32 *)d_g, (u32)half, d1, d2, top);
        else hipLaunchKernelGGL((ipa_collapse_wide<FQ>), grid, block, 0, st, (u32 *)d_g, (u32)half, d1, d2, top);
    } else {
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_collapse<FP>), grid, block, 0, st, (u32 *)d_g, (u32)half, d1, d2, top);
        else hipLaunchKernelGGL((ipa_collapse<FQ>), grid, block, 0, st, (u32 *)d_g, (u32)half, d1, d2, top);
    }
    if (form == H2_FORM_CANONICAL) {
        dim3 g2((unsigned)((half * 2 + 255) / 256));
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_to_mont<FP>), g2, block, 0, st, (u32 *)d_g, half * 2, 0);
        else hipLaunchKernelGGL((ipa_to_mont<FQ>), g2, block, 0, st, (u32 *)d_g, half * 2, 0);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

static int round_scalars_launch(int field, const void *d_p, unsigned k, unsigned j, const u64 *challenges, int form, void *d_cl,
                                void *d_cr, hipStream_t st) {
    IpaContext &cx = g_ipa_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    u64 um[32 * 4];
    for (unsigned r = 0; r < j; ++r) host_to_mont(field, um + 4 * r, challenges + 4 * r, form);
    int rc = cx.stage.reserve(32 * 32);
    if (rc == H2_OK) rc = cx.stab.reserve(((size_t)32 << j));
    if (rc != H2_OK) return rc;
    // stream-ordered after the previous round's kernels that read the staging buffer; pageable source consumed on return
    if (j) H2_HIP(hipMemcpyAsync(cx.stage.ptr, um, 32 * j, hipMemcpyHostToDevice, st));
    dim3 block(256), gs((unsigned)((((size_t)1 << j) + 255) / 256)), gc((unsigned)((((size_t)1 << k) + 255) / 256));
    if (field == H2_FP) {
        hipLaunchKernelGGL((ipa_s_table<FP>), gs, block, 0, st, cx.stab.as<u32>(), cx.stage.as<u32>(), j);
        hipLaunchKernelGGL((ipa_round_scalars<FP>), gc, block, 0, st, (const u32 *)d_p, cx.stab.as<u32>(), k, j, (u32 *)d_cl, (u32 *)d_cr);
    } else {
        hipLaunchKernelGGL((ipa_s_table<FQ>), gs, block, 0, st, cx.stab.as<u32>(), cx.stage.as<u32>(), j);
        hipLaunchKernelGGL((ipa_round_scalars<FQ>), gc, block, 0, st, (const u32 *)d_p, cx.stab.as<u32>(), k, j, (u32 *)d_cl, (u32 *)d_cr);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---- the round loop's own launches (h2_ipa_rounds_device): three per round beside the commit -----------------------------------
// A round used to enqueue ten small launches and a copy around its commit (two inner products of two launches each, the challenge
// upload, the s table, the round scalars, the tails, two folds); the rounds after the switch to the collapsed generators are chains
// of short launches, so each one costs as much as the work it carries.
static constexpr u32 kIpBlocks = 128;       // partial sums per inner product
template <int F> __device__ __forceinline__ fe ipa_block_sum(u32 *sh, fe v) {       // 256 lanes; the sum lands in every lane of wave 0's lane 0
    fe_store(sh + 8 * threadIdx.x, v);
    __syncthreads();
    for (u32 off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) fe_store(sh + 8 * threadIdx.x, fe_add<F>(fe_load(sh + 8 * threadIdx.x), fe_load(sh + 8 * (threadIdx.x + off))));
        __syncthreads();
    }
    return fe_load(sh);
}
// blocks [0, 2 kIpBlocks): partial sums of <p'_hi, b_lo> (first kIpBlocks) and <p'_lo, b_hi>, raw Montgomery products;
// the blocks after them: the round's scalars over the original generators (ipa_round_scalars' body; s = this round's table)
template <int F>
__global__ void __launch_bounds__(256) ipa_round_prep(const u32 *__restrict__ p, const u32 *__restrict__ b, const u32 *__restrict__ s, u32 k, u32 j,
                                                      u32 *__restrict__ partial, u32 *__restrict__ cl, u32 *__restrict__ cr) {
    __shared__ __attribute__((aligned(16))) u32 sh[256 * 8];
    const u32 blk = k - j, half = 1u << (blk - 1);
    if (blockIdx.x < 2 * kIpBlocks) {
        const u32 which = blockIdx.x / kIpBlocks, bi = blockIdx.x % kIpBlocks;
        const u32 *pa = p + (which ? 0 : 8 * (size_t)half), *pb = b + (which ? 8 * (size_t)half : 0);
        fe acc = fe_zero();
        for (u32 i = bi * 256 + threadIdx.x; i < half; i += kIpBlocks * 256)
            acc = fe_add<F>(acc, fe_mulx<F>(fe_load(pa + 8 * (size_t)i), fe_load(pb + 8 * (size_t)i)));
        acc = ipa_block_sum<F>(sh, acc);
        if (threadIdx.x == 0) fe_store(partial + 8 * (size_t)blockIdx.x, acc);
        return;
    }
    const u32 m = (blockIdx.x - 2 * kIpBlocks) * 256 + threadIdx.x;
    if (m >> k) return;
    const u32 h = m >> blk, i = m & ((1u << blk) - 1);
    const fe v = fe_mulx<F>(fe_load(p + 8 * (size_t)(i ^ half)), fe_load(s + 8 * (size_t)h));
    if (cl == cr) {
        fe_store(cl + 8 * (size_t)m, v);
        return;
    }
    const bool lo = i < half;
    fe_store(cl + 8 * (size_t)m, lo ? v : fe_zero());
    fe_store(cr + 8 * (size_t)m, lo ? fe_zero() : v);
}
// one workgroup: the two inner products from their partial sums, then the rows behind the generators' (ipa_round_tails)
template <int F>
__global__ void __launch_bounds__(256) ipa_round_finish(const u32 *__restrict__ partial, fe z, fe l_rand, fe r_rand, u32 *__restrict__ vl,
                                                        u32 *__restrict__ vr, u32 *__restrict__ bl, u32 *__restrict__ br) {
    __shared__ __attribute__((aligned(16))) u32 sh[256 * 8];
    // lanes 0..127 carry the first product's partial sums, 128..255 the second's: one tree, stopped one level early
    fe_store(sh + 8 * threadIdx.x, fe_load(partial + 8 * (size_t)threadIdx.x));
    __syncthreads();
    for (u32 off = 64; off > 0; off >>= 1) {
        const u32 g = threadIdx.x >> 7, l = threadIdx.x & 127;
        if (l < off) fe_store(sh + 8 * threadIdx.x, fe_add<F>(fe_load(sh + 8 * threadIdx.x), fe_load(sh + 8 * (g * 128 + l + off))));
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fe_store(vl, fe_mulx<F>(fe_load(sh), z));
        fe_store(bl, l_rand);
    } else if (threadIdx.x == 128) {
        fe_store(vr, fe_mulx<F>(fe_load(sh + 8 * 128), z));
        fe_store(br, r_rand);
    }
}
// blocks [0, fold_blocks): p'[i] += p'[i + half] u^-1 and b[i] += b[i + half] u (prover.rs:128-133); the blocks after them: the next
// round's s table, s_{j+1}(h) = s_j(h >> 1) * u^{h & 1} (the definition at ipa_s_table), from one buffer into the other
template <int F>
__global__ void __launch_bounds__(256) ipa_round_fold(u32 *__restrict__ p, u32 *__restrict__ b, u32 half, u32 fold_blocks, fe u_inv, fe u,
                                                      const u32 *__restrict__ s_old, u32 *__restrict__ s_new, u32 s_count) {
    if (blockIdx.x < fold_blocks) {
        const u32 i = blockIdx.x * 256 + threadIdx.x;
        if (i >= half) return;
        fe_store(p + 8 * (size_t)i, fe_add<F>(fe_load(p + 8 * (size_t)i), fe_mulx<F>(fe_load(p + 8 * (size_t)(half + i)), u_inv)));
        fe_store(b + 8 * (size_t)i, fe_add<F>(fe_load(b + 8 * (size_t)i), fe_mulx<F>(fe_load(b + 8 * (size_t)(half + i)), u)));
        return;
    }
    const u32 h = (blockIdx.x - fold_blocks) * 256 + threadIdx.x;
    if (h >= s_count) return;
    const fe v = fe_load(s_old + 8 * (size_t)(h >> 1));
    fe_store(s_new + 8 * (size_t)h, (h & 1) ? fe_mulx<F>(v, u) : v);
}

// a[0] -= *v: the constant coefficient of s_poly / p' after their evaluation at x_3 (prover.rs:51, :72), without a host round trip
template <int F> __global__ void __launch_bounds__(64) ipa_sub_at0(u32 *__restrict__ a, const u32 *__restrict__ v) {
    if (blockIdx.x == 0 && threadIdx.x == 0) fe_store(a, fe_sub<F>(fe_load(a), fe_load(v)));
}

// s[0] -= ev[0] + xp1 ev[1] + xp2 ev[2] + xp3 ev[3]: the evaluation of s_poly at x_3 from the evaluations of its four quarters (each a polynomial in the
// quarter's own index; xp_r = x_3^(r n / 4)), subtracted from the constant coefficient (prover.rs:49-51) -- one lane
template <int F> __global__ void __launch_bounds__(64) ipa_fix_s0(u32 *__restrict__ s0, const u32 *__restrict__ ev, fe xp1, fe xp2, fe xp3) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    fe t = fe_load(ev);
    t = fe_add<F>(t, fe_mulx<F>(fe_load(ev + 8), xp1));
    t = fe_add<F>(t, fe_mulx<F>(fe_load(ev + 16), xp2));
    t = fe_add<F>(t, fe_mulx<F>(fe_load(ev + 24), xp3));
    fe_store(s0, fe_sub<F>(fe_load(s0), t));
}

}  // namespace h2

using namespace h2;

extern "C" int h2_generator_collapse_device(int curve, void *d_g_xy, size_t half, const uint64_t *challenge, int form, void *stream) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !challenge ||
        (half && !d_g_xy) || half > (1u << 30))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!half) return H2_OK;
    return collapse_launch(curve, d_g_xy, half, challenge, form, (hipStream_t)stream);
}

extern "C" int h2_generator_collapse(int curve, uint64_t *g_xy, size_t half, const uint64_t *challenge, int form) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !challenge ||
        (half && !g_xy) || half > (1u << 30))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!half) return H2_OK;
    void *d = nullptr;
    H2_HIP(hipMalloc(&d, half * 128));
    hipError_t e = hipMemcpy(d, g_xy, half * 128, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = collapse_launch(curve, d, half, challenge, form, 0);
        if (rc == H2_OK) e = hipMemcpy(g_xy, d, half * 64, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}

static int fold_launch(int field, void *d_a, size_t half, const u64 *factor, int form, hipStream_t st) {
    u64 fm[4];
    host_to_mont(field, fm, factor, form);     // a Montgomery factor works for data in either form
    fe f;
    memcpy(f.v, fm, 32);
    dim3 grid((unsigned)((half + 255) / 256)), block(256);
    if (field == H2_FP) hipLaunchKernelGGL((ipa_fold<FP>), grid, block, 0, st, (u32 *)d_a, (u32)half, f);
    else hipLaunchKernelGGL((ipa_fold<FQ>), grid, block, 0, st, (u32 *)d_a, (u32)half, f);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_fold_scalars_device(int field, void *d_a, size_t half, const uint64_t *factor, int form, void *stream) {
    if ((field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !factor || (half && !d_a) ||
        half > (1u << 30))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!half) return H2_OK;
    return fold_launch(field, d_a, half, factor, form, (hipStream_t)stream);
}

extern "C" int h2_fold_scalars(int field, uint64_t *a, size_t half, const uint64_t *factor, int form) {
    if ((field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !factor || (half && !a) ||
        half > (1u << 30))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!half) return H2_OK;
    void *d = nullptr;
    H2_HIP(hipMalloc(&d, half * 64));
    hipError_t e = hipMemcpy(d, a, half * 64, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = fold_launch(field, d, half, factor, form, 0);
        if (rc == H2_OK) e = hipMemcpy(a, d, half * 32, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}

extern "C" int h2_ipa_round_scalars_device(int field, const void *d_p, unsigned k, unsigned j, const uint64_t *challenges, int form,
                                           void *d_cl, void *d_cr, void *stream) {
    if ((field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || k < 1 || k > 30 || j >= k ||
        (j && !challenges) || !d_p || !d_cl || !d_cr)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return round_scalars_launch(field, d_p, k, j, challenges, form, d_cl, d_cr, (hipStream_t)stream);
}


// `rounds` rounds of the loop below over ONE registered basis (the public entry switches bases in between)
static int ipa_rounds_impl(int curve, unsigned k, unsigned rounds, h2_bases_t basis, int paired, void *d_p, void *d_b, const uint64_t *z,
                           const uint64_t *rands, void *d_column_l, void *d_column_r, h2_ipa_write_point_fn write_point,
                           h2_ipa_squeeze_fn squeeze, void *user, uint64_t *challenges_out, uint64_t *c_out, uint64_t *f_acc, void *stream) {
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int sf = curve == H2_PALLAS ? H2_FQ : H2_FP, bf = curve == H2_PALLAS ? H2_FP : H2_FQ;
    const size_t n = (size_t)1 << k;
    IpaContext &cx = g_ipa_ctxs.get(st);           // the caller holds cx.rounds_mu
    if ((rc = cx.rounds.reserve(64 + 192 + (size_t)2 * kIpBlocks * 32)) != H2_OK) return rc;
    const size_t tab_words = (size_t)8 << (rounds ? rounds - 1 : 0);          // s_j has 2^j entries, j < rounds
    if ((rc = cx.rstab.reserve(2 * tab_words * 4)) != H2_OK) return rc;         // two tables: a round reads one, its fold writes the next
    if (!cx.rounds_host) H2_HIP(hipHostMalloc(&cx.rounds_host, 256, hipHostMallocDefault));
    u32 *d_ip = cx.rounds.as<u32>(), *d_lr = d_ip + 16, *d_partial = d_ip + 64;
    u32 *stab[2] = {cx.rstab.as<u32>(), cx.rstab.as<u32>() + tab_words};
    H2_HIP(hipMemcpyAsync(stab[0], kHostField[sf].one, 32, hipMemcpyHostToDevice, st));      // s_0 = [1]
    u64 *lr = (u64 *)cx.rounds_host;
    u64 challenges[32 * 4];
    u32 *col_l = (u32 *)d_column_l, *col_r = paired ? col_l : (u32 *)d_column_r;
    fe zf;
    memcpy(zf.v, z, 32);
    for (unsigned j = 0; j < rounds; ++j) {
        const size_t half = (size_t)1 << (k - j - 1);
        fe lf, rf;
        memcpy(lf.v, rands + 8 * j, 32);
        memcpy(rf.v, rands + 8 * j + 4, 32);
        u32 *vl = col_l + 8 * n, *vr = paired ? col_l + 8 * (n + 1) : col_r + 8 * n;
        u32 *bl = paired ? col_l + 8 * (n + 2) : col_l + 8 * (n + 1), *br = paired ? col_l + 8 * (n + 3) : col_r + 8 * (n + 1);
        // both inner products' partial sums and the round's scalars in one launch, the sums and the tail rows in a second
        const dim3 gp(2 * kIpBlocks + (unsigned)((n + 255) / 256));
        if (sf == H2_FP) {
            hipLaunchKernelGGL((ipa_round_prep<FP>), gp, dim3(256), 0, st, (const u32 *)d_p, (const u32 *)d_b, (const u32 *)stab[j & 1], k, j, d_partial, col_l, col_r);
            hipLaunchKernelGGL((ipa_round_finish<FP>), dim3(1), dim3(256), 0, st, (const u32 *)d_partial, zf, lf, rf, vl, vr, bl, br);
        } else {
            hipLaunchKernelGGL((ipa_round_prep<FQ>), gp, dim3(256), 0, st, (const u32 *)d_p, (const u32 *)d_b, (const u32 *)stab[j & 1], k, j, d_partial, col_l, col_r);
            hipLaunchKernelGGL((ipa_round_finish<FQ>), dim3(1), dim3(256), 0, st, (const u32 *)d_partial, zf, lf, rf, vl, vr, bl, br);
        }
        H2_HIP(hipGetLastError());
        if (paired) {
            rc = h2_commit_pair_device(basis, col_l, n + 4, k - j - 1, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_lr, st);
        } else {
            const void *cols[2] = {col_l, col_r};
            void *outs[2] = {d_lr, d_lr + 24};
            rc = h2_commit_batch_device(basis, cols, 2, n + 2, nullptr, nullptr, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, outs, st);
        }
        if (rc != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(lr, d_lr, 192, hipMemcpyDeviceToHost, st));
        H2_HIP(hipStreamSynchronize(st));
        // .to_affine() of both points with one inversion (prover.rs:116-117), then the transcript (:121-124)
        const u64 *zl = lr + 8, *zr = lr + 12 + 8;
        if (host_is_zero(zl) || host_is_zero(zr)) {
            set_last_error_msg("h2_ipa_rounds_device: L_j or R_j is the point at infinity, which a transcript cannot absorb");
            return H2_ERR_ARGS;
        }
        u64 zz[4], zi[4], zli[4], zri[4];
        host_mul(bf, zz, zl, zr);
        host_inv(bf, zi, zz);
        host_mul(bf, zli, zi, zr);
        host_mul(bf, zri, zi, zl);
        for (int s = 0; s < 2; ++s) {
            const u64 *pt = lr + 12 * s, *inv = s ? zri : zli;
            u64 i2[4], i3[4], xy[8];
            host_mul(bf, i2, inv, inv);
            host_mul(bf, i3, i2, inv);
            host_mul(bf, xy, pt, i2);
            host_mul(bf, xy + 4, pt + 4, i3);
            if ((rc = write_point(user, xy)) != H2_OK) return rc;
        }
        u64 *u = challenges + 4 * j, u_inv[4], t[4];
        if ((rc = squeeze(user, u)) != H2_OK) return rc;
        if (host_is_zero(u)) {
            set_last_error_msg("h2_ipa_rounds_device: zero challenge");       // the reference unwraps the inverse (:125)
            return H2_ERR_ARGS;
        }
        host_inv(sf, u_inv, u);
        {   // the two folds (:128-133) and the next round's s table in one launch
            fe uf, uif;
            memcpy(uf.v, u, 32);
            memcpy(uif.v, u_inv, 32);
            const u32 fold_blocks = (u32)((half + 255) / 256), s_count = j + 1 < rounds ? 2u << j : 0u;
            const dim3 gf(fold_blocks + (s_count + 255) / 256);
            if (sf == H2_FP)
                hipLaunchKernelGGL((ipa_round_fold<FP>), gf, dim3(256), 0, st, (u32 *)d_p, (u32 *)d_b, (u32)half, fold_blocks, uif, uf, (const u32 *)stab[j & 1],
                                   stab[(j + 1) & 1], s_count);
            else
                hipLaunchKernelGGL((ipa_round_fold<FQ>), gf, dim3(256), 0, st, (u32 *)d_p, (u32 *)d_b, (u32)half, fold_blocks, uif, uf, (const u32 *)stab[j & 1],
                                   stab[(j + 1) & 1], s_count);
            H2_HIP(hipGetLastError());
        }
        host_mul(sf, t, rands + 8 * j, u_inv);                                                          // :140-141
        host_add(sf, f_acc, f_acc, t);
        host_mul(sf, t, rands + 8 * j + 4, u);
        host_add(sf, f_acc, f_acc, t);
    }
    if (rounds == k) {                   // p' has collapsed to the scalar c (:146)
        H2_HIP(hipMemcpyAsync(lr, d_p, 32, hipMemcpyDeviceToHost, st));
        H2_HIP(hipStreamSynchronize(st));
        memcpy(c_out, lr, 32);
    }
    if (challenges_out) memcpy(challenges_out, challenges, 32 * (size_t)rounds);
    return H2_OK;
}

// The round loop of `commitment::create_proof` (poly/commitment/prover.rs:104-142) as ONE call: per round the inner products,
// the L_j / R_j scalars over the original generators, their commit(s), the two points to the transcript, the challenge, and the
// p' / b folds.  The host is touched once per round (192 bytes of L_j, R_j in Jacobian form land in pinned memory; the
// normalisation -- one shared inversion -- and the challenge's inverse are microseconds of 64-bit host arithmetic), through
// the caller's transcript: `write_point` receives the affine point (8 x u64, Montgomery), `squeeze` returns the challenge
// scalar (4 x u64, Montgomery) -- the two TranscriptWrite methods the reference's loop uses (:121-124).
// With switch_rounds = J > 0 the first J rounds run over `basis` (the original generators), then G'_J is read off its table
// (h2_ipa_collapsed_generators_device), registered as a table of its own next to u and w, and the remaining k - J rounds run as a
// (k - J)-round argument over it: a round over the original generators costs a full-size commit whatever j, a round over G'_J a
// commit of 2^(k-J) points.
// Measured (profiles/r04_opening_switch_sweep.txt, ms per argument): k = 16: J = 2 8.8 / 1 9.3; k = 18: J = 4 10.7 / 3 11.6; k = 19: 5 13.4 / 4 13.6;
// k = 20: 5 19.5 / 6 20.0 / 4 21.2; k = 21: 6 31.8 / 5 32.2; k = 22: 5 58.2 / 6 59.8 / 4 61.5 / 8 66.1 -- a table of 2^14 points up to k = 19, then
// five rounds: every further round over the original generators costs a full 2^k commit, a larger G' only its registration.
// Round 5: rounds over a table of up to 2^16 points are paired commits with 8-bit sub-digits (msm.hip, pair_subdigit_launch: 0.20 ms at 2^14
// points against 0.245 at 2^15 and 0.26 for the general form at either), which moves k = 20 to a 2^14-point table as well: J = 6 16.8 ms, J = 5
// 16.9-17.4, the general form 17.6-17.7 either way (profiles/r05_pair_subdigits.txt).  k = 21 keeps 6 rounds, k >= 22 five.
extern "C" unsigned h2_ipa_default_switch_rounds(unsigned k, int paired) {
    if (!paired || k < 16 || k > 26 || h2_commit_window_bits(((size_t)1 << k) + 4) != 16) return 0;
    return k <= 20 ? k - 14 : k == 21 ? 6 : 5;
}

extern "C" int h2_ipa_rounds_device(int curve, unsigned k, unsigned switch_rounds, h2_bases_t basis, int paired, void *d_p, void *d_b,
                                    const uint64_t *z, const uint64_t *rands, const uint64_t *uw_xy, void *d_column_l, void *d_column_r,
                                    h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze, void *user, uint64_t *c_out,
                                    uint64_t *f_out, void *stream) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || k < 1 || k > 30 || !d_p || !d_b || !z || !rands || !d_column_l || !write_point ||
        !squeeze || !c_out || !f_out || (!paired && !d_column_r))
        return H2_ERR_ARGS;
    // everything that can be refused is refused HERE, before round 0 writes L_0 / R_0 into the caller's transcript and folds p' / b:
    // the read-out of the collapsed generators (h2_ipa_collapsed_generators_device) takes a 16-bit table over g || u || u || w || w
    size_t basis_n = 0;
    int basis_c = 0, basis_curve = -1;
    if ((h2_bases_info(basis, &basis_n, &basis_c, &basis_curve)) != H2_OK) return H2_ERR_HANDLE;
    if (basis_curve != curve || basis_n < ((size_t)1 << k)) return H2_ERR_ARGS;
    const bool can_switch = paired && basis_c == 16 && k <= 26 && basis_n == ((size_t)1 << k) + 4;
    unsigned J = switch_rounds == H2_IPA_SWITCH_DEFAULT ? (can_switch ? h2_ipa_default_switch_rounds(k, paired) : 0) : switch_rounds;
    if (J && (!can_switch || J >= k || J > 12 || !uw_xy)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    IpaContext &cx = g_ipa_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.rounds_mu);
    u64 f_acc[4] = {0, 0, 0, 0};
    if (!J) {
        rc = ipa_rounds_impl(curve, k, k, basis, paired, d_p, d_b, z, rands, d_column_l, d_column_r, write_point, squeeze, user, nullptr,
                             c_out, f_acc, stream);
        if (rc == H2_OK) memcpy(f_out, f_acc, 32);
        return rc;
    }
    u64 challenges[12 * 4];
    if ((rc = ipa_rounds_impl(curve, k, J, basis, 1, d_p, d_b, z, rands, d_column_l, nullptr, write_point, squeeze, user, challenges, c_out,
                              f_acc, stream)) != H2_OK)
        return rc;
    const unsigned kj = k - J;
    const size_t nj = (size_t)1 << kj;
    const bool pair2 = h2_commit_pair_supported(nj + 4) != 0;
    const size_t tail = pair2 ? 4 : 2;
    if ((rc = cx.gprime.reserve((nj + tail) * 64)) != H2_OK) return rc;
    char *d_g = cx.gprime.as<char>();
    if ((rc = h2_ipa_collapsed_generators_device(basis, k, J, challenges, H2_FORM_MONTGOMERY, d_g, st)) != H2_OK) return rc;
    u64 tails[4 * 8];                                                 // u, u, w, w  or  u, w
    for (size_t t = 0; t < tail; ++t) memcpy(tails + 8 * t, uw_xy + 8 * (pair2 ? t / 2 : t), 64);
    H2_HIP(hipMemcpyAsync(d_g + nj * 64, tails, tail * 64, hipMemcpyHostToDevice, st));
    H2_HIP(hipStreamSynchronize(st));                                // the registration reads the points on the null stream
    static const bool keep_table = [] { const char *e = ab_env("H2_IPA_KEEP_TABLE"); return !(e && e[0] == '0'); }();      // 0: register / free per argument (A/B)
    if (cx.gp_handle && (!keep_table || cx.gp_curve != curve || cx.gp_n != nj + tail)) {
        (void)h2_bases_free(cx.gp_handle);
        cx.gp_handle = 0;
    }
    if (cx.gp_handle) {
        if ((rc = bases_refill_device(cx.gp_handle, d_g, nj + tail, H2_FORM_MONTGOMERY)) != H2_OK) return rc;
    } else {
        // rounds over a small table are sub-digit paired commits, which read an ENDOMORPHISM table as well: 8 x 16 doublings in its chain instead
        // of 15 x 16 -- the chain is what the table costs (0.77 ms of 240 dependent doublings at 2^14 points).  H2_IPA_GLV_TABLE=0: the plain table (A/B).
        static const bool glv_env = [] { const char *e = ab_env("H2_IPA_GLV_TABLE"); return !(e && e[0] == '0'); }();
        const bool glv = glv_env && pair2 && pair_subdigits_apply(nj + tail);
        if ((rc = bases_register_device_internal(curve, d_g, nj + tail, H2_FORM_MONTGOMERY, &cx.gp_handle, glv)) != H2_OK) return rc;
        cx.gp_curve = curve;
        cx.gp_n = nj + tail;
    }
    const h2_bases_t hj = cx.gp_handle;
    // the columns of the second phase fit in the first phase's scratch (2 (nj + 2) <= 2^k + 4)
    void *col_l = d_column_l, *col_r = pair2 ? nullptr : (void *)((char *)d_column_l + 32 * (nj + 2));
    rc = ipa_rounds_impl(curve, kj, kj, hj, pair2 ? 1 : 0, d_p, d_b, z, rands + 8 * J, col_l, col_r, write_point, squeeze, user, nullptr,
                         c_out, f_acc, stream);
    if (rc == H2_OK) memcpy(f_out, f_acc, 32);
    // Retention: the table (16 rows x (nj + tail) x 64 B) stays with this (device, stream) context for the next argument of the same
    // shape -- up to 2^17 points (128 MiB); beyond that, and with H2_IPA_KEEP_TABLE=0 (the A/B arm: register / free per argument), it
    // goes back at the END of the call.  h2_trim releases what is kept.  (ipa_rounds_impl synchronised the stream before returning
    // c and f to the host, so nothing is reading the table any more.)
    if (!keep_table || nj + tail > ((size_t)1 << 17) + 4) {
        (void)h2_bases_free(cx.gp_handle);
        cx.gp_handle = 0;
    }
    return rc;
}

// the same with p' and b in host memory (copied in; the folded vectors are not copied back: only c and f leave the argument)
extern "C" int h2_ipa_rounds(int curve, unsigned k, unsigned switch_rounds, h2_bases_t basis, int paired, const uint64_t *p, const uint64_t *b,
                             const uint64_t *z, const uint64_t *rands, const uint64_t *uw_xy, h2_ipa_write_point_fn write_point,
                             h2_ipa_squeeze_fn squeeze, void *user, uint64_t *c_out, uint64_t *f_out) {
    if (k < 1 || k > 30 || !p || !b) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    const size_t n = (size_t)1 << k, col = (n + 4) * 32;
    char *d = nullptr;
    H2_HIP(hipMalloc((void **)&d, 2 * n * 32 + 2 * col));
    hipError_t e = hipMemcpy(d, p, n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n * 32, b, n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        rc = h2_ipa_rounds_device(curve, k, switch_rounds, basis, paired, d, d + n * 32, z, rands, uw_xy, d + 2 * n * 32, d + 2 * n * 32 + col,
                                  write_point, squeeze, user, c_out, f_out, nullptr);
    (void)hipFree(d);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}

// ---- the whole opening argument: poly::commitment::prover::create_proof (prover.rs:26-151) as ONE call -----------------------------
// What a Rust shim replaces is the function, not its loop: the caller draws the randomness (n + 1 + 2k scalars in the reference's
// order, :45-47, :53, :111-112) and owns the transcript; everything between is here.  With the round loop alone native
// (h2_ipa_rounds) the C++ host mirror spent 68-71 ms on a k = 20 argument whose device work is 18: two host Horner evaluations
// of 2^20 coefficients and five 32 MiB vectors across PCIe between host-pointer calls.  Here the vectors cross once (h2_open) or
// not at all (h2_open_device), the constant-coefficient corrections (:51, :72) are one-lane kernels instead of host round trips,
// and what does not depend on xi -- b = powers of x_3 (:86-97) and v -- runs while the host normalises and hashes the S
// commitment.  v: P'(x_3) = xi s(x_3) + p(x_3) with s(x_3) = 0 EXACTLY after :51 (field arithmetic), so v = p(x_3) is
// evaluated before xi exists and p' is never read a second time.  Same bytes as the reference for the same randomness
// (tests/test_gpu_opening.py).
static int open_impl(IpaContext &cx, int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                     const uint64_t *uw_xy, const void *d_p, const uint64_t *host_p, const uint64_t *p_blind, const uint64_t *x3, void *d_s,
                     const uint64_t *host_s, const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze,
                     void *user, uint64_t *c_out, uint64_t *f_out, hipStream_t st) {
    const int sf = curve == H2_PALLAS ? H2_FQ : H2_FP, bf = curve == H2_PALLAS ? H2_FP : H2_FQ;
    const size_t n = (size_t)1 << k;
    int rc;
    if ((rc = cx.open_b.reserve(n * 32)) != H2_OK || (rc = cx.open_col.reserve((2 * n + 8) * 32)) != H2_OK || (rc = cx.open_small.reserve(1024)) != H2_OK) return rc;
    if (!cx.open_host) H2_HIP(hipHostMalloc(&cx.open_host, 256, hipHostMallocDefault));
    if (!cx.open_ev) H2_HIP(hipEventCreateWithFlags(&cx.open_ev, hipEventDisableTiming));
    if (!cx.open_ev2) H2_HIP(hipEventCreateWithFlags(&cx.open_ev2, hipEventDisableTiming));
    if (!cx.open_side) H2_HIP(hipStreamCreateWithFlags(&cx.open_side, hipStreamNonBlocking));
    u32 *small = cx.open_small.as<u32>(), *d_s_at = small, *d_v = small + 8, *d_commit = small + 16, *d_blind = small + 40;
    u32 *d_ev = small + 64, *d_part = small + 128;              // (ranges) the quarters' evaluations, 4 x 8 words; their partial commitments, 4 x 24 words
    void *d_b = cx.open_b.ptr;
    u64 *land = (u64 *)cx.open_host;
    // s_poly gets its root at x_3 (:49-51) and is committed to (:56)
    static const bool ranges_env = [] { const char *e = ab_env("H2_OPEN_S_RANGES"); return !(e && e[0] == '0'); }();      // 0: one upload, one commit (A/B)
    // (with p_poly in host memory too -- h2_open -- the calling thread is the bottleneck either way: 96 MiB of pageable copies through one thread, and every launch
    // between two of them is PCIe idle time; measured there the quarters LOSE a millisecond to one upload + one commit, so they serve the resident p_poly only)
    const bool by_ranges = host_s && !host_p && ranges_env && k >= 16;
    if (by_ranges) {
        // The fresh coefficients come from the caller's rng, i.e. from HOST memory: 32 MiB across PCIe at k = 20 (0.9 ms) in front of a 1.1 ms commit.
        // Cut into four quarters that cross from the top down: quarter r's share of the commitment (h2_commit_range_device over its columns of the
        // table) starts as soon as it has landed, on one of two side streams, while the next quarter crosses; every quarter is evaluated at x_3 on the
        // way (as a polynomial in its own index).  The quarter with the constant coefficient crosses LAST: s(x_3) = sum_r x_3^(r n / 4) ev_r is known a
        // kernel later, the coefficient is fixed, that quarter is committed with the blind, and the four shares are added.  Behind the last byte: one
        // quarter-size commit instead of a whole one.
        const size_t q = n / 4;
        if (!cx.open_side2) H2_HIP(hipStreamCreateWithFlags(&cx.open_side2, hipStreamNonBlocking));
        for (hipEvent_t *e : {&cx.open_land[0], &cx.open_land[1], &cx.open_land[2], &cx.open_land[3], &cx.open_fixed, &cx.open_parts})
            if (!*e) H2_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        H2_HIP(hipMemcpyAsync(d_blind, s_blind, 32, hipMemcpyHostToDevice, st));
        u64 xq[4], xp[4][4];                                    // x_3^(n / 4) by squaring, then its first three powers
        memcpy(xq, x3, 32);
        for (unsigned i = 0; i + 2 < k; ++i) host_mul(sf, xq, xq, xq);
        memcpy(xp[1], xq, 32);
        host_mul(sf, xp[2], xq, xq);
        host_mul(sf, xp[3], xp[2], xq);
        hipStream_t sides[2] = {cx.open_side, cx.open_side2};
        for (int r = 3; r >= 0; --r) {
            char *dst = (char *)d_s + (size_t)r * q * 32;
            H2_HIP(hipMemcpyAsync(dst, (const char *)host_s + (size_t)r * q * 32, q * 32, hipMemcpyHostToDevice, st));
            if ((rc = h2_eval_polynomial_device(sf, dst, q, x3, H2_FORM_MONTGOMERY, d_ev + 8 * r, st)) != H2_OK) return rc;
            if (r == 0) {
                fe f1, f2, f3;
                memcpy(f1.v, xp[1], 32);
                memcpy(f2.v, xp[2], 32);
                memcpy(f3.v, xp[3], 32);
                if (sf == H2_FP) hipLaunchKernelGGL((ipa_fix_s0<FP>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_ev, f1, f2, f3);
                else hipLaunchKernelGGL((ipa_fix_s0<FQ>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_ev, f1, f2, f3);
                H2_HIP(hipGetLastError());
            }
            H2_HIP(hipEventRecord(cx.open_land[r], st));
            hipStream_t sd = sides[r & 1];
            H2_HIP(hipStreamWaitEvent(sd, cx.open_land[r], 0));
            if ((rc = h2_commit_range_device(g_basis, dst, (size_t)r * q, q, r == 0 ? d_blind : nullptr, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_part + 24 * r,
                                             sd)) != H2_OK)
                return rc;
        }
        // the four shares meet on the first side stream (quarter 0 ran on it last; the other stream's two are awaited)
        H2_HIP(hipEventRecord(cx.open_parts, sides[1]));
        H2_HIP(hipStreamWaitEvent(sides[0], cx.open_parts, 0));
        if ((rc = h2_points_sum_device(curve, d_part, 4, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_commit, sides[0])) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(land, d_commit, 96, hipMemcpyDeviceToHost, sides[0]));
        H2_HIP(hipEventRecord(cx.open_ev, sides[0]));
    } else {
        if (!host_p) {
            // a resident p_poly may have been produced on the caller's stream: the side stream (b and v, below) waits for what is on `st` NOW --
            // recorded before the evaluation and the commitment of s_poly are enqueued, so that it runs beside them, not behind them
            H2_HIP(hipEventRecord(cx.open_ev2, st));
            H2_HIP(hipStreamWaitEvent(cx.open_side, cx.open_ev2, 0));
        }
        if (host_s) H2_HIP(hipMemcpyAsync(d_s, host_s, n * 32, hipMemcpyHostToDevice, st));
        if ((rc = h2_eval_polynomial_device(sf, d_s, n, x3, H2_FORM_MONTGOMERY, d_s_at, st)) != H2_OK) return rc;
        if (sf == H2_FP) hipLaunchKernelGGL((ipa_sub_at0<FP>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_s_at);
        else hipLaunchKernelGGL((ipa_sub_at0<FQ>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_s_at);
        H2_HIP(hipGetLastError());
        H2_HIP(hipMemcpyAsync(d_blind, s_blind, 32, hipMemcpyHostToDevice, st));
        if ((rc = h2_commit_device(g_basis, d_s, n, nullptr, d_blind, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_commit, st)) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(land, d_commit, 96, hipMemcpyDeviceToHost, st));
        H2_HIP(hipEventRecord(cx.open_ev, st));
    }
    // beside the commit and the host's part: p_poly's way in (host vectors), b and v -- on the side stream, or (S commitment by ranges: the side streams
    // carry its shares) on `st` itself, which has only moved and evaluated the quarters so far
    hipStream_t side = by_ranges ? st : cx.open_side;
    if (host_p) {
        if ((rc = cx.open_p.reserve(n * 32)) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(cx.open_p.ptr, host_p, n * 32, hipMemcpyHostToDevice, side));
        d_p = cx.open_p.ptr;
    }
    if ((rc = h2_powers_device(sf, x3, n, H2_FORM_MONTGOMERY, d_b, side)) != H2_OK) return rc;
    if ((rc = h2_eval_polynomial_device(sf, d_p, n, x3, H2_FORM_MONTGOMERY, d_v, side)) != H2_OK) return rc;
    H2_HIP(hipEventRecord(cx.open_ev2, side));
    H2_HIP(hipEventSynchronize(cx.open_ev));
    {   // .to_affine() (:56), the transcript (:57), xi (:62) and z (:66)
        const u64 *Z = land + 8;
        if (host_is_zero(Z)) {
            (void)hipStreamSynchronize(side);
            set_last_error_msg("h2_open: the commitment to s_poly is the point at infinity, which a transcript cannot absorb");
            return H2_ERR_ARGS;
        }
        u64 zi[4], i2[4], i3[4], xy[8];
        host_inv(bf, zi, Z);
        host_mul(bf, i2, zi, zi);
        host_mul(bf, i3, i2, zi);
        host_mul(bf, xy, land, i2);
        host_mul(bf, xy + 4, land + 4, i3);
        if ((rc = write_point(user, xy)) != H2_OK) { (void)hipStreamSynchronize(side); return rc; }
    }
    u64 xi[4], z[4];
    if ((rc = squeeze(user, xi)) != H2_OK || (rc = squeeze(user, z)) != H2_OK) { (void)hipStreamSynchronize(side); return rc; }
    // P' = P - [v] G_0 + [xi] S (:70-73), in place on s_poly
    H2_HIP(hipStreamWaitEvent(st, cx.open_ev2, 0));
    if ((rc = h2_scale_add_device(sf, d_s, xi, d_p, n, H2_FORM_MONTGOMERY, st)) != H2_OK) return rc;
    if (sf == H2_FP) hipLaunchKernelGGL((ipa_sub_at0<FP>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_v);
    else hipLaunchKernelGGL((ipa_sub_at0<FQ>), dim3(1), dim3(64), 0, st, (u32 *)d_s, (const u32 *)d_v);
    H2_HIP(hipGetLastError());
    u64 f0[4], t[4], f_delta[4];
    host_mul(sf, t, s_blind, xi);                                                                      // :74-78
    host_add(sf, f0, t, p_blind);
    char *col = cx.open_col.as<char>();
    rc = h2_ipa_rounds_device(curve, k, switch_rounds, opening_basis, paired, d_s, d_b, z, rands, uw_xy, col, paired ? nullptr : col + (n + 4) * 32,
                              write_point, squeeze, user, c_out, f_delta, st);
    if (rc != H2_OK) { (void)hipStreamSynchronize(side); return rc; }
    host_add(sf, f_out, f0, f_delta);
    return H2_OK;
}

static bool open_bad_args(int curve, unsigned k, const uint64_t *p_blind, const uint64_t *x3, const uint64_t *s_blind, const uint64_t *rands,
                          h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze, uint64_t *c_out, uint64_t *f_out) {
    return (curve != H2_PALLAS && curve != H2_VESTA) || k < 1 || k > 30 || !p_blind || !x3 || !s_blind || !rands || !write_point || !squeeze ||
           !c_out || !f_out;
}

// what can be refused is refused before the S commitment reaches the caller's transcript
static int open_check_bases(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired) {
    size_t gn = 0, on = 0;
    int gc = -1, oc = -1;
    if (h2_bases_info(g_basis, &gn, nullptr, &gc) != H2_OK || h2_bases_info(opening_basis, &on, nullptr, &oc) != H2_OK) return H2_ERR_HANDLE;
    const size_t n = (size_t)1 << k;
    if (gc != curve || oc != curve || gn != n || on != n + (paired ? 4 : 2)) return H2_ERR_ARGS;
    if (h2_bases_blind_base_set(g_basis) != 1) return H2_ERR_ARGS;             // Params::w must be installed (h2_bases_set_blind_base)
    return H2_OK;
}

extern "C" int h2_open_device(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                              const uint64_t *uw_xy, const void *d_p_poly, const uint64_t *p_blind, const uint64_t *x3, void *d_s_poly,
                              const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze,
                              void *user, uint64_t *c_out, uint64_t *f_out, void *stream) {
    if (open_bad_args(curve, k, p_blind, x3, s_blind, rands, write_point, squeeze, c_out, f_out) || !d_p_poly || !d_s_poly ||
        d_p_poly == d_s_poly)                          // (s_poly is overwritten with P' while p_poly is still being read)
        return H2_ERR_ARGS;
    int rc = open_check_bases(curve, k, g_basis, opening_basis, paired);
    if (rc != H2_OK) return rc;
    if ((rc = ensure_device()) != H2_OK) return rc;
    IpaContext &cx = g_ipa_ctxs.get((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(cx.open_mu);
    return open_impl(cx, curve, k, g_basis, opening_basis, paired, switch_rounds, uw_xy, d_p_poly, nullptr, p_blind, x3, d_s_poly, nullptr, s_blind, rands,
                     write_point, squeeze, user, c_out, f_out, (hipStream_t)stream);
}

// p_poly resident, the fresh s_poly where a host rng leaves it: its quarters cross PCIe inside the call and are committed as they land (open_impl)
extern "C" int h2_open_device_host_s(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                                     const uint64_t *uw_xy, const void *d_p_poly, const uint64_t *p_blind, const uint64_t *x3, const uint64_t *s_poly,
                                     const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze,
                                     void *user, uint64_t *c_out, uint64_t *f_out, void *stream) {
    if (open_bad_args(curve, k, p_blind, x3, s_blind, rands, write_point, squeeze, c_out, f_out) || !d_p_poly || !s_poly) return H2_ERR_ARGS;
    int rc = open_check_bases(curve, k, g_basis, opening_basis, paired);
    if (rc != H2_OK) return rc;
    if ((rc = ensure_device()) != H2_OK) return rc;
    IpaContext &cx = g_ipa_ctxs.get((hipStream_t)stream);
    std::lock_guard<std::mutex> lk(cx.open_mu);
    if ((rc = cx.open_s.reserve(((size_t)1 << k) * 32)) != H2_OK) return rc;
    return open_impl(cx, curve, k, g_basis, opening_basis, paired, switch_rounds, uw_xy, d_p_poly, nullptr, p_blind, x3, cx.open_s.ptr, s_poly, s_blind, rands,
                     write_point, squeeze, user, c_out, f_out, (hipStream_t)stream);
}

// the same from host vectors (what `&Polynomial<C::Scalar, Coeff>` and a Vec of fresh randomness are): both cross PCIe once, p_poly
// beside the commitment to s_poly; nothing is copied back
extern "C" int h2_open(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                       const uint64_t *uw_xy, const uint64_t *p_poly, const uint64_t *p_blind, const uint64_t *x3, const uint64_t *s_poly,
                       const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze, void *user,
                       uint64_t *c_out, uint64_t *f_out) {
    if (open_bad_args(curve, k, p_blind, x3, s_blind, rands, write_point, squeeze, c_out, f_out) || !p_poly || !s_poly) return H2_ERR_ARGS;
    int rc = open_check_bases(curve, k, g_basis, opening_basis, paired);
    if (rc != H2_OK) return rc;
    if ((rc = ensure_device()) != H2_OK) return rc;
    IpaContext &cx = g_ipa_ctxs.get(nullptr);
    std::lock_guard<std::mutex> lk(cx.open_mu);
    const size_t n = (size_t)1 << k;
    if ((rc = cx.open_s.reserve(n * 32)) != H2_OK) return rc;
    return open_impl(cx, curve, k, g_basis, opening_basis, paired, switch_rounds, uw_xy, nullptr, p_poly, p_blind, x3, cx.open_s.ptr, s_poly, s_blind, rands,
                     write_point, squeeze, user, c_out, f_out, nullptr);
}
