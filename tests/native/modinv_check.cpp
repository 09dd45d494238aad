// CPU check of halo2_amd/csrc/field_inv.cuh (the divstep inversion the device uses): the same functions compiled for the host, against
// inverses computed by Fermat with 128-bit arithmetic here.  g++ -O2 -std=c++17 tests/native/modinv_check.cpp -o build/modinv_check
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "../../halo2_amd/csrc/field_inv.cuh"
#include "../../halo2_amd/csrc/host_field.h"      // host_inv (the divsteps above, Montgomery in and out) and the binary Euclid it replaced

typedef unsigned __int128 u128;
struct U256 { uint64_t w[4]; };
static const uint64_t P_FP[4] = {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL};
static const uint64_t P_FQ[4] = {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL};
static bool geq(const uint64_t *a, const uint64_t *b) { for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i]; return true; }
static void sub(uint64_t *a, const uint64_t *b) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 t = (u128)a[i] - b[i] - br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } }
// (a * b) mod p by shift-and-add (slow, obviously right)
static U256 mulmod(const U256 &a, const U256 &b, const uint64_t *p) {
    U256 r = {{0, 0, 0, 0}};
    for (int i = 255; i >= 0; --i) {
        uint64_t c = r.w[3] >> 63;                                  // r = 2 r mod p   (r < p < 2^255: no overflow beyond bit 255)
        for (int k = 3; k > 0; --k) r.w[k] = (r.w[k] << 1) | (r.w[k - 1] >> 63);
        r.w[0] <<= 1;
        (void)c;
        if (geq(r.w, p)) sub(r.w, p);
        if ((b.w[i >> 6] >> (i & 63)) & 1) {
            u128 cy = 0;
            for (int k = 0; k < 4; ++k) { u128 t = (u128)r.w[k] + a.w[k] + cy; r.w[k] = (uint64_t)t; cy = t >> 64; }
            if (geq(r.w, p)) sub(r.w, p);
        }
    }
    return r;
}
int main() {
    std::mt19937_64 rng(0x5AFE6CD);
    int bad = 0, done = 0;
    for (int field = 0; field < 2; ++field) {
        const uint64_t *p = field ? P_FQ : P_FP;
        uint32_t pw[8];
        memcpy(pw, p, 32);
        for (int it = 0; it < 600; ++it) {
            U256 x;
            for (int k = 0; k < 4; ++k) x.w[k] = rng();
            x.w[3] &= 0x3FFFFFFFFFFFFFFFULL;                          // < 2^254 < p
            if (it == 0) x = {{0, 0, 0, 0}};
            if (it == 1) x = {{1, 0, 0, 0}};
            if (it == 2) { memcpy(x.w, p, 32); x.w[0] -= 1; }        // p - 1
            if (it == 3) x = {{2, 0, 0, 0}};
            if (it == 4) x = {{0, 0, 0, 0x2000000000000000ULL}};     // a power of two
            if (it >= 5 && it < 40) { x = {{0, 0, 0, 0}}; x.w[(it - 5) / 9] = 1ULL << (7 * ((it - 5) % 9)); }
            uint32_t xw[8], ow[8];
            memcpy(xw, x.w, 32);
            h2::modinv30(xw, pw, ow);
            U256 inv;
            memcpy(inv.w, ow, 32);
            const bool zero = !(x.w[0] | x.w[1] | x.w[2] | x.w[3]);
            const U256 prod = mulmod(x, inv, p);
            const bool ok = zero ? !(inv.w[0] | inv.w[1] | inv.w[2] | inv.w[3])
                                 : (prod.w[0] == 1 && !(prod.w[1] | prod.w[2] | prod.w[3]) && !geq(inv.w, p));
            if (!ok) { if (bad < 5) printf("FAIL field %d case %d\n", field, it); ++bad; }
            ++done;
        }
    }
    printf("modinv30: %d cases, %d failures\n", done, bad);
    // the library's host inversion (csrc/host_field.h: what the opening argument's round loop and the transcript call twice per round) against
    // the binary extended Euclid it replaced and against a * a^-1 = 1 in Montgomery form; raw limbs at or above p (up to 2^256 - 1) included
    int hbad = 0, hdone = 0;
    for (int field = 0; field < 2; ++field) {
        for (int it = 0; it < 4000; ++it) {
            uint64_t a[4], r1[4], r2[4], prod[4];
            for (int k = 0; k < 4; ++k) a[k] = rng();
            if (it % 3) a[3] &= 0x3FFFFFFFFFFFFFFFULL;
            if (it == 0) memset(a, 0, 32);
            if (it == 1) memcpy(a, h2::kHostField[field].p, 32);                          // p itself: zero
            if (it == 2) memcpy(a, h2::kHostField[field].one, 32);
            h2::host_inv(field, r1, a);
            h2::host_inv_euclid(field, r2, a);
            bool ok = memcmp(r1, r2, 32) == 0;
            h2::host_mul(field, prod, a, r1);
            uint64_t red[4];
            memcpy(red, a, 32);
            while (h2::host_ge(red, h2::kHostField[field].p)) h2::host_sub_raw(red, red, h2::kHostField[field].p);
            const bool zero = !(red[0] | red[1] | red[2] | red[3]);
            ok = ok && (zero ? !(r1[0] | r1[1] | r1[2] | r1[3]) : memcmp(prod, h2::kHostField[field].one, 32) == 0);
            if (!ok) { if (hbad < 5) printf("FAIL host_inv field %d case %d\n", field, it); ++hbad; }
            ++hdone;
        }
    }
    printf("host_inv: %d cases, %d failures\n", hdone, hbad);
    return bad || hbad ? 1 : 0;
}
