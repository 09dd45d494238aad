// Modular inversion by divsteps ("safegcd": D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion",
// TCHES 2019(3)) on signed 30-bit limbs: x^-1 mod p for the two Pasta moduli, constant time, no data-dependent branch -- what a SIMD
// lane wants.  Replaces the Fermat ladder a^(p-2) (256 squarings + ~80 products = ~84 000 instructions on the 8 x 32 layer, all on the
// calling lane's dependent chain) wherever one lane has to invert: the normalisation of a point (xyzz_to_affine), the shared inversion
// of a batch (poly_batch_invert, msm_table_normalise_batch), hash-to-curve.
//
// The iteration (section 11 of the paper, in its delta = 1/2 form): state (zeta, f, g) with f odd; a divstep maps
//     g odd, zeta < 0 :  (zeta, f, g) -> (-zeta - 2, g, (g + f) / 2)      [written below as: conditionally f += g after g += f]
//     g odd, zeta >= 0:  (zeta, f, g) -> ( zeta - 1, f, (g - f) / 2)      [with x = -f]
//     g even          :  (zeta, f, g) -> ( zeta - 1, f,  g      / 2)
// 590 divsteps bring any (f, g) with f odd, 0 <= g <= f < 2^256 to g = 0, f = +-gcd; 20 batches of 30 are run.  A batch looks at the low
// limbs only and yields a 2 x 2 integer matrix t with t (f, g) = 2^30 (f', g'); the same matrix is applied to (d, e), the running
// cofactors with d x = f, e x = g (mod p), the division by 2^30 being made exact by adding the right multiple of p (p = 1 mod 2^30
// for both fields, so that multiple is read off the low limb).  At the end f = +-1 and x^-1 = +-d.
//
// Cost, counted on the generated ISA (hipcc -S of a kernel that calls fe_inv): 1042 instructions per batch of 30 divsteps with its two matrix
// applications (98 of them 64-bit multiply-adds), 20 batches, 50 + 668 around them (the repacking, the normalisation, the two products with
// R^2): 21 558 instructions per inversion = 131 multiplications of the carry-free layer (165 instructions each) against ~510 for the ladder.
// Measured (profiles/r05_inversion_ab.txt): h2_batch_invert at 2^20 0.250 -> 0.134 ms, a 2^12-point multiexp with affine output 0.685 -> 0.518 ms.
// (The round-4 review asked whether a constant-time inversion lands under ~80 multiplication-equivalents, the point at which per-lane
// batched-affine bucket additions would start to pay: it does not -- DESIGN.md section 4.3.)
//
// Plain integer code, __host__ __device__: tests/native/modinv_check.cpp runs the very same functions on the CPU against big-integer
// inverses (no GPU needed); on the device they are checked through every affine output of the parity suite.
#pragma once
#include <stdint.h>

#ifndef H2_HD
#if defined(__HIPCC__) || defined(__CUDACC__)
#define H2_HD __host__ __device__ __forceinline__
#else
#define H2_HD inline
#endif
#endif

namespace h2 {

struct s30 {
    int32_t v[9];          // value = sum v[i] 2^(30 i); limbs 0..7 in [0, 2^30) after an update, limb 8 signed
};
struct t2x2 {
    int32_t u, v, q, r;    // t = [u v; q r], entries in (-2^30, 2^30]
};
static constexpr int32_t kM30 = (int32_t)(0xFFFFFFFFu >> 2);

// 30 divsteps on the low limbs of f and g; returns the new zeta
H2_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, t2x2 &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);            // all ones iff zeta < 0
        const uint32_t c2 = 0u - (g & 1u);               // all ones iff g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;      // +-(f, u, v): minus when zeta >= 0 ... see the note below
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                                        // zeta < 0 and g odd: the swap case
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;       // -zeta - 2, or zeta - 1
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}
// (note on signs: with c1 the mask of zeta < 0, x = (f ^ c1) - c1 = -f when zeta < 0 and +f otherwise, i.e. the code keeps the NEGATED
// convention of the paper's g - f / g + f cases; the swap step f += g then restores f = old g.  What matters -- and what the CPU test checks
// against big integers -- is the invariant t (f, g) = 2^30 (f', g') with f' odd and the final f = +-1.)

// (d, e) <- t (d, e) / 2^30 mod p, limbs renormalised; d, e stay in (-2p, p)
H2_HD void update_de_30(s30 &d, s30 &e, const t2x2 &t, const s30 &p) {
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);          // + p for every negative input, so the result cannot drift below -2p
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0], ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    // p = 1 mod 2^30: the multiple of p that clears the low 30 bits is minus those bits
    md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)kM30);
    me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)kM30);
    cd += (int64_t)p.v[0] * md;
    ce += (int64_t)p.v[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)p.v[i] * md;
        ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)p.v[i] * me;
        d.v[i - 1] = (int32_t)cd & kM30;
        e.v[i - 1] = (int32_t)ce & kM30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
// (f, g) <- t (f, g) / 2^30 (exact)
H2_HD void update_fg_30(s30 &f, s30 &g, const t2x2 &t) {
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0], cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
        cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
        f.v[i - 1] = (int32_t)cf & kM30;
        g.v[i - 1] = (int32_t)cg & kM30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// r in (-2p, p) -> sign * r in [0, p)   (sign < 0: negate)
H2_HD void normalize_30(s30 &r, int32_t sign, const s30 &p) {
    int32_t add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] += p.v[i] & add;
    const int32_t neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (r.v[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= kM30;
    }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] += p.v[i] & add;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= kM30;
    }
}
// 8 x 32-bit words (value < 2^256) <-> nine 30-bit limbs
H2_HD s30 s30_from_words(const uint32_t w[8]) {
    s30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, k = bit >> 5, s = bit & 31;
        uint32_t lo = w[k] >> s;
        if (s > 2 && k + 1 < 8) lo |= w[k + 1] << (32 - s);
        r.v[i] = (int32_t)(lo & (uint32_t)kM30);
    }
    return r;
}
H2_HD void s30_to_words(const s30 &a, uint32_t w[8]) {      // a normalised, in [0, 2^256)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int lo = 30 * i - 32 * k;
            if (lo > -30 && lo < 32) x |= lo >= 0 ? (uint32_t)a.v[i] << lo : (uint32_t)a.v[i] >> (-lo);
        }
        w[k] = x;
    }
}
// out = x^-1 mod p as integers (0 for x = 0), x < p, p odd with p = 1 mod 2^30, p < 2^256
H2_HD void modinv30(const uint32_t x[8], const uint32_t p_words[8], uint32_t out[8]) {
    const s30 p = s30_from_words(p_words);
    s30 d = {{0, 0, 0, 0, 0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}}, f = p, g = s30_from_words(x);
    int32_t zeta = -1;
    for (int i = 0; i < 20; ++i) {               // 600 >= 590 divsteps
        t2x2 t;
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de_30(d, e, t, p);
        update_fg_30(f, g, t);
    }
    normalize_30(d, f.v[8], p);                  // f = +-1: its sign is the sign of its top limb
    s30_to_words(d, out);
}

}  // namespace h2
