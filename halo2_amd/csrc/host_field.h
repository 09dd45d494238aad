// Host-side Pasta field arithmetic for the handful of constants a call needs (twiddle steps, fused scale factors,
// challenge recoding).  4 x 64-bit Montgomery limbs, R = 2^256 -- same values as field.cuh's device constants.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/halo2_mi355x.h"
#include "field_inv.cuh"

namespace h2 {

typedef uint64_t u64;
typedef unsigned __int128 u128;
struct HostField {
    u64 p[4], inv, r2[4], one[4];
};
static const HostField kHostField[2] = {
    {{0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL}, 0x992d30ecffffffffULL,
     {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL},
     {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
    {{0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL}, 0x8c46eb20ffffffffULL,
     {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL},
     {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
};
static void host_mul(int f, u64 r[4], const u64 a[4], const u64 b[4]) {
    const HostField &F = kHostField[f];
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * F.inv;
        c = ((u128)m * F.p[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F.p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) {
        ge = true;
        for (int i = 3; i >= 0; i--) {
            if (t[i] > F.p[i]) break;
            if (t[i] < F.p[i]) { ge = false; break; }
        }
    }
    if (ge) {
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)t[i] - F.p[i] - (u64)br;
            t[i] = (u64)d;
            br = (d >> 64) & 1;
        }
    }
    memcpy(r, t, 32);
}
// bring a caller-supplied constant into Montgomery form
static void host_to_mont(int f, u64 r[4], const u64 *a, int form) {
    if (form == H2_FORM_MONTGOMERY) memcpy(r, a, 32);
    else host_mul(f, r, a, kHostField[f].r2);
}

// Montgomery -> canonical
static inline void host_from_mont(int f, u64 r[4], const u64 a[4]) {
    const u64 one[4] = {1, 0, 0, 0};
    host_mul(f, r, a, one);
}

static inline void host_add(int f, u64 r[4], const u64 a[4], const u64 b[4]) {
    const HostField &F = kHostField[f];
    u64 t[4];
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a[i] + b[i];
        t[i] = (u64)c;
        c >>= 64;
    }
    bool ge = true;                       // both below p < 2^255: no carry out of 256 bits
    for (int i = 3; i >= 0; --i) {
        if (t[i] > F.p[i]) break;
        if (t[i] < F.p[i]) { ge = false; break; }
    }
    if (ge) {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)t[i] - F.p[i] - (u64)br;
            t[i] = (u64)d;
            br = (d >> 64) & 1;
        }
    }
    memcpy(r, t, 32);
}

// 256-bit helpers of the inversion below
static inline bool host_ge(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
static inline u64 host_sub_raw(u64 r[4], const u64 a[4], const u64 b[4]) {        // returns the borrow
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - (u64)br;
        r[i] = (u64)d;
        br = (d >> 64) & 1;
    }
    return (u64)br;
}
static inline void host_halve_mod(const u64 p[4], u64 x[4]) {                      // x / 2 mod p (p odd), x < p
    u64 carry = 0;
    if (x[0] & 1) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)x[i] + p[i];
            x[i] = (u64)c;
            c >>= 64;
        }
        carry = (u64)c;
    }
    for (int i = 0; i < 4; ++i) x[i] = (x[i] >> 1) | ((i < 3 ? x[i + 1] : carry) << 63);
}
static inline void host_shr1(u64 x[4]) {
    for (int i = 0; i < 4; ++i) x[i] = (x[i] >> 1) | ((i < 3 ? x[i + 1] : 0) << 63);
}

// 1 / a, Montgomery in and out (a != 0; 0 -> 0): the divstep inversion of field_inv.cuh on the raw limbs -- the very functions the device
// runs, here on the host: 2.9 us against 9.3 us for the binary extended Euclid below (one Xeon core; 20 000 random inputs agree, also
// raw limbs at or above p) -- then two products by R^2 to land back in Montgomery form: (a R)^-1 = a^-1 R^-1 -> a^-1 R.  The opening
// argument inverts twice per round on the host (the shared Z of L_j, R_j and the challenge: poly/commitment/prover.rs:116-117, :125).
static inline void host_inv_euclid(int f, u64 r[4], const u64 a[4]);
static inline void host_inv(int f, u64 r[4], const u64 a[4]) {
    const HostField &F = kHostField[f];
    u64 u[4], o[4], t[4];
    memcpy(u, a, 32);
    while (host_ge(u, F.p)) host_sub_raw(u, u, F.p);                  // raw limbs at or above p (at most 3 p < 2^256): reduce first
    if (!(u[0] | u[1] | u[2] | u[3])) {                               // 0 has no inverse: return 0 (as ff's `invert().unwrap_or(0)` callers do)
        memset(r, 0, 32);
        return;
    }
    uint32_t uw[8], pw[8], ow[8];                                     // little-endian host, as everywhere in this library
    memcpy(uw, u, 32);
    memcpy(pw, F.p, 32);
    modinv30(uw, pw, ow);
    memcpy(o, ow, 32);
    host_mul(f, t, o, F.r2);
    host_mul(f, r, t, F.r2);
}
// the binary extended Euclid host_inv used through round 4 (~500 shift / subtract steps); kept as the cross-check of the one above
// (tests/native/modinv_check.cpp)
static inline void host_inv_euclid(int f, u64 r[4], const u64 a[4]) {
    const HostField &F = kHostField[f];
    u64 u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    memcpy(u, a, 32);
    memcpy(v, F.p, 32);
    while (host_ge(u, v)) host_sub_raw(u, u, v);                     // raw limbs at or above p (at most 3 p < 2^256): reduce first
    if (!(u[0] | u[1] | u[2] | u[3])) {                               // 0 has no inverse: return 0 (as ff's `invert().unwrap_or(0)`
        memset(r, 0, 32);                                             // callers do) instead of halving 0 for ever
        return;
    }
    auto is_one = [](const u64 t[4]) { return t[0] == 1 && !(t[1] | t[2] | t[3]); };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1)) {
            host_shr1(u);
            host_halve_mod(F.p, x1);
        }
        while (!(v[0] & 1)) {
            host_shr1(v);
            host_halve_mod(F.p, x2);
        }
        if (host_ge(u, v)) {
            host_sub_raw(u, u, v);
            if (host_sub_raw(x1, x1, x2)) {            // x1 - x2 mod p
                u128 c = 0;
                for (int i = 0; i < 4; ++i) {
                    c += (u128)x1[i] + F.p[i];
                    x1[i] = (u64)c;
                    c >>= 64;
                }
            }
        } else {
            host_sub_raw(v, v, u);
            if (host_sub_raw(x2, x2, x1)) {
                u128 c = 0;
                for (int i = 0; i < 4; ++i) {
                    c += (u128)x2[i] + F.p[i];
                    x2[i] = (u64)c;
                    c >>= 64;
                }
            }
        }
    }
    u64 t[4];
    host_mul(f, t, is_one(u) ? x1 : x2, F.r2);
    host_mul(f, r, t, F.r2);
}

static inline bool host_is_zero(const u64 a[4]) { return !(a[0] | a[1] | a[2] | a[3]); }

}  // namespace h2
