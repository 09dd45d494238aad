// Pasta hash-to-curve on the device: `C::CurveExt::hash_to_curve(domain_prefix)(message)` as `Params::new` uses it
// (halo2_proofs/src/poly/commitment.rs:52-62, :102-104: 2^k + 2 points per Params; minutes of CPU time at k = 20 in the
// reference).  The function lives in the dependency pasta_curves 0.5.1; this is its published construction (Zcash protocol
// specification 5.4.9.8; RFC 9380 expand_message_xmd / hash_to_field / simplified SWU), one lane per message:
//
//   DST' = prefix || "-" || curve || "_XMD:BLAKE2b_SSWU_RO_" || len        BLAKE2b-512, personalisation 16 zero bytes
//   b0 = H(0^128 || msg || 0x00 0x80 0x00 || DST'),  b1 = H(b0 || 0x01 || DST'),  b2 = H((b0 ^ b1) || 0x02 || DST')
//   u_i = OS2IP(b_(i+1)) mod p                      64 big-endian bytes -> lo R^2 + hi R^3 in Montgomery arithmetic
//   Q_i = map_to_curve_simple_swu(u_i)              on y^2 = x^3 + a_iso x + 1265, Z = -13; sgn0(y) = sgn0(u)
//   result = iso_map(Q_0 + Q_1)                     the 3-isogeny onto y^2 = x^3 + 5, derived in gen_h2c_consts.py
//
// Results are bit-exact with the reference: tests reproduce the verifying key pinned in halo2_proofs/tests/plonk_api.rs:958-981
// from `Params.new(5)` alone.  Arithmetic is the 8 x 32 layer (field.cuh): ~2.4e3 modular multiplications per point
// (four Fermat inversions, up to four square-root attempts), 2^20 points in tens of milliseconds.
#include <cstring>
#include <vector>

#include "common.h"
#include "field_sqrt.cuh"

namespace h2 {

#include "h2c_consts.inc"

typedef unsigned long long u64l;

struct H2cDst {
    unsigned char b[96];     // DST' = DST || len(DST)
    u32 len;
};

__device__ __forceinline__ u64l rotr64(u64l x, int n) { return (x >> n) | (x << (64 - n)); }

// BLAKE2b-512 of `len` bytes delivered by get(i); unkeyed, personalisation all zero
template <typename Get> __device__ void blake2b_512(Get get, u32 len, unsigned char out[64]) {
    const u64l IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    const unsigned char SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    u64l h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ULL;       // digest 64, key 0, fanout 1, depth 1
    const u32 nblocks = len == 0 ? 1 : (len + 127) / 128;
    for (u32 blk = 0; blk < nblocks; ++blk) {
        u64l m[16];
        for (int w = 0; w < 16; w++) {
            u64l x = 0;
            for (int k = 0; k < 8; k++) {
                const u32 idx = blk * 128 + w * 8 + k;
                if (idx < len) x |= (u64l)get(idx) << (8 * k);
            }
            m[w] = x;
        }
        const bool last = blk + 1 == nblocks;
        const u64l t = last ? (u64l)len : (u64l)(blk + 1) * 128;
        u64l v[16];
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        for (int r = 0; r < 12; r++) {
            const unsigned char *s = SIGMA[r];
#define H2_B2G(a, b, c, d, x, y)                                     \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);        \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);        \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);        \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);
            H2_B2G(0, 4, 8, 12, m[s[0]], m[s[1]])
            H2_B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
            H2_B2G(2, 6, 10, 14, m[s[4]], m[s[5]])
            H2_B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
            H2_B2G(0, 5, 10, 15, m[s[8]], m[s[9]])
            H2_B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
            H2_B2G(2, 7, 8, 13, m[s[12]], m[s[13]])
            H2_B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
#undef H2_B2G
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 8; k++) out[8 * i + k] = (unsigned char)(h[i] >> (8 * k));
}

// 64 big-endian bytes -> field element (Montgomery): from_uniform_bytes of the reversed string
template <int F> __device__ fe h2c_field_from_be64(const unsigned char d[64]) {
    fe lo, hi;
    for (int j = 0; j < 8; j++) {
        lo.v[j] = (u32)d[63 - 4 * j] | (u32)d[62 - 4 * j] << 8 | (u32)d[61 - 4 * j] << 16 | (u32)d[60 - 4 * j] << 24;
        hi.v[j] = (u32)d[31 - 4 * j] | (u32)d[30 - 4 * j] << 8 | (u32)d[29 - 4 * j] << 16 | (u32)d[28 - 4 * j] << 24;
    }
    return fe_add<F>(fe_mulx<F>(lo, fe_r2<F>()), fe_mulx<F>(hi, h2c_r3<F>()));
}

template <int F> __device__ __forceinline__ u32 fe_sgn0(const fe &a_mont) { return fe_from_mont<F>(a_mont).v[0] & 1u; }

template <int F> __device__ __forceinline__ fe iso_rhs(const fe &x) {      // x^3 + a x + b on the iso curve
    return fe_add<F>(fe_mulx<F>(fe_add<F>(fe_sqr<F>(x), h2c_iso_a<F>()), x), h2c_iso_b<F>());
}

// RFC 9380 6.6.2 (AB != 0)
template <int F> __device__ void map_to_curve_simple_swu(const fe &u, fe &x, fe &y) {
    const fe zu2 = fe_mulx<F>(h2c_swu_z<F>(), fe_sqr<F>(u));
    const fe tv = fe_add<F>(fe_sqr<F>(zu2), zu2);
    fe x1;
    if (fe_is_zero(tv)) x1 = h2c_b_over_za<F>();
    else x1 = fe_mulx<F>(h2c_neg_b_over_a<F>(), fe_add<F>(fe_one<F>(), fe_inv<F>(tv)));
    x = x1;
    if (!fe_sqrt<F>(iso_rhs<F>(x1), y)) {
        x = fe_mulx<F>(zu2, x1);
        (void)fe_sqrt<F>(iso_rhs<F>(x), y);          // one of the two is always a square
    }
    if (fe_sgn0<F>(u) != fe_sgn0<F>(y)) y = fe_neg<F>(y);
}

template <int F>
__global__ void __launch_bounds__(128) h2c_kernel(const unsigned char *__restrict__ msgs, u32 msg_len, size_t count, H2cDst dst, int out_mont,
                                                  u32 *__restrict__ out_xy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned char *msg = msgs + i * (size_t)msg_len;
    unsigned char b0[64], b1[64], b2[64];
    const u32 dl = dst.len;
    blake2b_512([&](u32 k) -> unsigned char {
        if (k < 128) return 0;
        k -= 128;
        if (k < msg_len) return msg[k];
        k -= msg_len;
        if (k < 3) return k == 1 ? 128 : 0;             // I2OSP(128, 2) || I2OSP(0, 1)
        return dst.b[k - 3];
    }, 128 + msg_len + 3 + dl, b0);
    blake2b_512([&](u32 k) -> unsigned char { return k < 64 ? b0[k] : k == 64 ? 1 : dst.b[k - 65]; }, 65 + dl, b1);
    blake2b_512([&](u32 k) -> unsigned char { return k < 64 ? (unsigned char)(b0[k] ^ b1[k]) : k == 64 ? 2 : dst.b[k - 65]; }, 65 + dl, b2);
    const fe u0 = h2c_field_from_be64<F>(b1), u1 = h2c_field_from_be64<F>(b2);
    fe x0, y0, x1, y1;
    map_to_curve_simple_swu<F>(u0, x0, y0);
    map_to_curve_simple_swu<F>(u1, x1, y1);
    // Q0 + Q1 on the iso curve (affine chord / tangent; the sum is the identity when Q1 = -Q0)
    fe ox = fe_zero(), oy = fe_zero();
    bool inf = false;
    fe lam;
    if (fe_eq(x0, x1)) {
        if (fe_eq(y0, y1) && !fe_is_zero(y0)) {
            const fe xx = fe_sqr<F>(x0);
            lam = fe_mulx<F>(fe_add<F>(fe_add<F>(fe_dbl<F>(xx), xx), h2c_iso_a<F>()), fe_inv<F>(fe_dbl<F>(y0)));
        } else {
            inf = true;
        }
    } else {
        lam = fe_mulx<F>(fe_sub<F>(y1, y0), fe_inv<F>(fe_sub<F>(x1, x0)));
    }
    if (!inf) {
        const fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), x0), x1);
        const fe y3 = fe_sub<F>(fe_mulx<F>(lam, fe_sub<F>(x0, x3)), y0);
        // iso_map (Velu, normalised): X = c^2 (x + t / d + u / d^2), Y = c^3 y (1 - t / d^2 - 2u / d^3), d = x - x0
        const fe d = fe_sub<F>(x3, h2c_iso_x0<F>());
        if (!fe_is_zero(d)) {                              // a kernel point maps to the identity
            const fe di = fe_inv<F>(d), di2 = fe_sqr<F>(di), di3 = fe_mulx<F>(di2, di);
            const fe X = fe_add<F>(x3, fe_add<F>(fe_mulx<F>(h2c_iso_t<F>(), di), fe_mulx<F>(h2c_iso_u<F>(), di2)));
            const fe Y = fe_mulx<F>(y3, fe_sub<F>(fe_sub<F>(fe_one<F>(), fe_mulx<F>(h2c_iso_t<F>(), di2)), fe_mulx<F>(h2c_iso_u2<F>(), di3)));
            ox = fe_mulx<F>(h2c_iso_c2<F>(), X);
            oy = fe_mulx<F>(h2c_iso_c3<F>(), Y);
        }
    }
    if (!out_mont) {
        ox = fe_from_mont<F>(ox);
        oy = fe_from_mont<F>(oy);
    }
    fe_store(out_xy + 16 * i, ox);
    fe_store(out_xy + 16 * i + 8, oy);
}

static int make_dst(int curve, const char *prefix, H2cDst &d) {
    const char *cid = curve == H2_PALLAS ? "pallas" : "vesta";
    const char *suffix = "_XMD:BLAKE2b_SSWU_RO_";
    const size_t lp = strlen(prefix), need = lp + 1 + strlen(cid) + strlen(suffix);
    if (lp > 64 || need + 1 > sizeof d.b) return H2_ERR_ARGS;
    size_t o = 0;
    memcpy(d.b + o, prefix, lp); o += lp;
    d.b[o++] = '-';
    memcpy(d.b + o, cid, strlen(cid)); o += strlen(cid);
    memcpy(d.b + o, suffix, strlen(suffix)); o += strlen(suffix);
    d.b[o++] = (unsigned char)need;
    d.len = (u32)o;
    return H2_OK;
}

}  // namespace h2

using namespace h2;

extern "C" int h2_hash_to_curve_device(int curve, const char *domain_prefix, const void *d_msgs, size_t msg_len, size_t count, int form,
                                       void *d_out_xy, void *stream) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !domain_prefix ||
        !d_out_xy || (count && msg_len && !d_msgs) || msg_len > 64 || count > ((size_t)1 << 31))
        return H2_ERR_ARGS;
    if (!count) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    H2cDst dst;
    memset(&dst, 0, sizeof dst);
    if ((rc = make_dst(curve, domain_prefix, dst)) != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((count + 127) / 128)), block(128);
    if (curve == H2_PALLAS)
        hipLaunchKernelGGL((h2c_kernel<FP>), grid, block, 0, st, (const unsigned char *)d_msgs, (u32)msg_len, count, dst, form == H2_FORM_MONTGOMERY, (u32 *)d_out_xy);
    else
        hipLaunchKernelGGL((h2c_kernel<FQ>), grid, block, 0, st, (const unsigned char *)d_msgs, (u32)msg_len, count, dst, form == H2_FORM_MONTGOMERY, (u32 *)d_out_xy);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_hash_to_curve(int curve, const char *domain_prefix, const uint8_t *msgs, size_t msg_len, size_t count, int form,
                                uint64_t *out_xy) {
    if (!out_xy || (count && msg_len && !msgs)) return H2_ERR_ARGS;
    if (!count) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    void *d_m = nullptr, *d_o = nullptr;
    hipError_t e = hipMalloc(&d_m, std::max<size_t>(count * msg_len, 16));
    if (e == hipSuccess) e = hipMalloc(&d_o, count * 64);
    if (e == hipSuccess && msg_len) e = hipMemcpy(d_m, msgs, count * msg_len, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = h2_hash_to_curve_device(curve, domain_prefix, d_m, msg_len, count, form, d_o, nullptr);
        if (rc == H2_OK) e = hipMemcpy(out_xy, d_o, count * 64, hipMemcpyDeviceToHost);
    }
    if (d_m) (void)hipFree(d_m);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) {
        set_last_hip_error(e, __FILE__, __LINE__);
        return H2_ERR_HIP;
    }
    return rc;
}
