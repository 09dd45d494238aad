// The Pasta curves' endomorphism phi(x, y) = (zeta x, y) = [lambda](x, y) (zeta, lambda: cube roots of unity of the base
// and scalar field) splits a 255-bit scalar k = k1 + k2 lambda with |k1|, |k2| < 2^128.  Used where a doubling chain is
// the latency floor: the generic multiexp (its Horner over windows shrinks from 255 to 128 doublings) and the
// opening argument's generator collapse (ipa.hip, split done once on the host there).
//
// Lattice basis (a1, b1), (a2, b2) with a + b lambda = 0 mod q (b1 < 0 for both fields) and g_i = floor(2^256 (b2, -b1) / q),
// all from extended Euclid on (q, lambda) in big-integer arithmetic offline; c_i = (k g_i) >> 256 only has to be CLOSE to the
// exact quotient -- any integers c1, c2 give k1 + k2 lambda = k -- closeness keeps k1, k2 short (max 128 bits over 20000
// random scalars per field; the digit code reserves 130).
#pragma once
#include "field.cuh"

namespace h2 {

template <int F> __device__ __forceinline__ fe glv_zeta() {   // base field F, Montgomery
    if (F == FP) return fe{{0x619a153du, 0x02021cf6u, 0x4980b78eu, 0x9e8c2697u, 0xc87a4666u, 0x2a676d5cu, 0xa7a17876u, 0x15d8049du}};
    return fe{{0x7feeeee3u, 0x410e7d20u, 0xd8fa2279u, 0x6afdf14fu, 0xeca4d4d7u, 0xfd3d8a04u, 0x77dba4efu, 0x2de2d607u}};
}

// out[0 .. NOUT) = limbs FIRST .. FIRST + NOUT of a * b (exact: all lower columns are summed for their carries)
template <int NA, int NB, int FIRST, int NOUT>
__device__ __forceinline__ void limbs_mul_window(const u32 *a, const u32 *b, u32 *out) {
    u64 acc = 0;
    u32 over = 0;   // carries out of the 64-bit column accumulator
#pragma unroll
    for (int col = 0; col < FIRST + NOUT; ++col) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = col - i;
            if (j >= 0 && j < NB) {
                const u64 p = (u64)a[i] * b[j];
                acc += p;
                over += acc < p;
            }
        }
        if (col >= FIRST) out[col - FIRST] = (u32)acc;
        acc = (acc >> 32) | ((u64)over << 32);
        over = 0;
    }
}

// r = r - s over 5 limbs (mod 2^160)
__device__ __forceinline__ void sub160(u32 *r, const u32 *s) {
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const u64 d = (u64)r[i] - s[i] - borrow;
        r[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
}
// two's complement 160-bit value -> magnitude; returns 1 when negative
__device__ __forceinline__ u32 abs160(u32 *v) {
    const u32 neg = v[4] >> 31;
    if (neg) {
        u32 carry = 1;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const u64 t = (u64)(~v[i]) + carry;
            v[i] = (u32)t;
            carry = (u32)(t >> 32);
        }
    }
    return neg;
}

// k: canonical scalar of field FS.  m1, m2: |k1|, |k2| (5 limbs, < 2^129); n1, n2: their signs (1 = negative)
template <int FS>
__device__ __forceinline__ void glv_split(const fe &k, u32 m1[5], u32 &n1, u32 m2[5], u32 &n2) {
    // FS = FQ: Pallas scalars; FS = FP: Vesta scalars
    const u32 a1[4] = {0x00000001u, FS == FQ ? 0x7fcae1c7u : 0x8cb12793u, FS == FQ ? 0x40f04915u : 0x40a89953u, 0x49e69d16u};
    const u32 b1[4] = {0x00000000u, FS == FQ ? 0x8cb12793u : 0x7fcae1c7u, FS == FQ ? 0x40a89953u : 0x40f04915u, 0x49e69d16u};   // |b1|
    const u32 a2[4] = {FS == FQ ? 0x00000000u : 0x00000001u, FS == FQ ? 0x8cb12793u : 0x0c7c095au, FS == FQ ? 0x40a89953u : 0x8198e269u,
                       FS == FQ ? 0x49e69d16u : 0x93cd3a2cu};
    const u32 b2[4] = {0x00000001u, FS == FQ ? 0x0c7c095au : 0x8cb12793u, FS == FQ ? 0x8198e269u : 0x40a89953u,
                       FS == FQ ? 0x93cd3a2cu : 0x49e69d16u};
    const u32 g1[5] = {FS == FQ ? 0x00000002u : 0x00000003u, FS == FQ ? 0x31f02568u : 0x32c49e4cu, FS == FQ ? 0x066389a4u : 0x02a2654eu,
                       FS == FQ ? 0x4f34e8b2u : 0x279a7459u, FS == FQ ? 0x00000002u : 0x00000001u};
    const u32 g2[5] = {0xffffffffu, FS == FQ ? 0x32c49e4bu : 0xff2b871bu, FS == FQ ? 0x02a2654eu : 0x03c12455u, 0x279a7459u, 0x00000001u};
    u32 c1[5], c2[5], t[5];
    limbs_mul_window<8, 5, 8, 5>(k.v, g1, c1);      // (k g1) >> 256
    limbs_mul_window<8, 5, 8, 5>(k.v, g2, c2);
#pragma unroll
    for (int i = 0; i < 5; ++i) m1[i] = k.v[i];      // k1 = k - c1 a1 - c2 a2   (mod 2^160: the true value has < 130 bits)
    limbs_mul_window<5, 4, 0, 5>(c1, a1, t);
    sub160(m1, t);
    limbs_mul_window<5, 4, 0, 5>(c2, a2, t);
    sub160(m1, t);
    limbs_mul_window<5, 4, 0, 5>(c1, b1, m2);        // k2 = c1 |b1| - c2 b2
    limbs_mul_window<5, 4, 0, 5>(c2, b2, t);
    sub160(m2, t);
    n1 = abs160(m1);
    n2 = abs160(m2);
}

}  // namespace h2
