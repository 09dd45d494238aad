// FFT over curve points: the Lagrange basis of a `Params`.
//
// Replaces the `best_fft::<Scalar, C::Curve>` call and the 2^-k scaling in `Params::new`
// (halo2_proofs/src/poly/commitment.rs:77-100):
//     g_lagrange[j] = 2^-k * sum_i alpha_inv^(i*j) * g[i],   alpha_inv = ROOT_OF_UNITY_INV^(2^(S-k)),
// returned as affine points (the batch_normalize of :90-100).  The result is a vector of group elements, so the
// butterfly order is free; this runs the same radix-2 DIT network as the field NTT, one kernel per stage over
// XYZZ points in HBM (128 B each): every butterfly is a full 255-bit scalar multiplication by its twiddle, so the
// transform is (n/2) log n scalar multiplications -- 10.5 M at k = 20 -- and nothing but VALU work.  Twiddles
// come from the field NTT's cached table.  One-off set-up work per Params (minutes on the CPU path).
#include "common.h"
#include "curve.cuh"
#include "glv.cuh"
#include "host_field.h"

namespace h2 {

__device__ __forceinline__ u32 ec_limb_at(const fe &s, int idx) {
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v = (idx == i) ? s.v[i] : v;
    return v;
}

// [k] p for a per-lane canonical scalar k (MSB-first double-and-add; lanes diverge on the adds)
template <int FB> __device__ xyzz<FB> ec_scalar_mul(const xyzz<FB> &p, const fe &k) {
    int top = -1;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (k.v[i]) top = 32 * i + 31 - __clz(k.v[i]);
    xyzz<FB> r = xyzz_identity<FB>();
    for (int b = top; b >= 0; --b) {
        r = xyzz_dbl<FB>(r);
        if ((ec_limb_at(k, b >> 5) >> (b & 31)) & 1) xyzz_add<FB>(r, p);
    }
    return r;
}

__device__ __forceinline__ u32 ec_bitrev(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// p[x] = g[bitrev(x)] as XYZZ
template <int FB>
__global__ void __launch_bounds__(256) ec_load_bitrev(const u32 *__restrict__ g, u32 *__restrict__ p, u32 n, int L, int mont) {
    u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    affine<FB> a = aff_load<FB>(g + 16 * (size_t)ec_bitrev(x, L));
    if (!mont) { a.x = fe_to_mont<FB>(a.x); a.y = fe_to_mont<FB>(a.y); }
    xyzz<FB> r = xyzz_identity<FB>();
    xyzz_madd<FB>(r, a);
    xyzz_store<FB>(p + 32 * (size_t)x, r);
}

// one DIT stage: pairs (x0, x0 + 2^t), twiddle omega^((x0 mod 2^t) * 2^(L-t-1))
template <int FB, int FS>
__global__ void __launch_bounds__(256) ec_stage(u32 *__restrict__ p, const u32 *__restrict__ tw, u32 half_n, int L, int t) {
    u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= half_n) return;
    const u32 low = q & ((1u << t) - 1);
    const u32 x0 = ((q >> t) << (t + 1)) | low, x1 = x0 + (1u << t);
    const size_t e = (size_t)low << (L - t - 1);
    xyzz<FB> a = xyzz_load<FB>(p + 32 * (size_t)x0), b = xyzz_load<FB>(p + 32 * (size_t)x1);
    if (e != 0) {
        fe w = fe_from_mont<FS>(fe_load(tw + 8 * e));
        b = ec_scalar_mul<FB>(b, w);
    }
    xyzz<FB> s = a, d = a;
    xyzz_add<FB>(s, b);
    b.y = fe_neg<FB>(b.y);
    xyzz_add<FB>(d, b);
    xyzz_store<FB>(p + 32 * (size_t)x0, s);
    xyzz_store<FB>(p + 32 * (size_t)x1, d);
}

// The same stage with one twiddle per WAVE (stages where a twiddle serves >= 64 butterflies, t <= L - 7): the 64 lanes of a
// wave take butterflies (hi, low) with a common `low`, so the scalar is wave-uniform and the walk has no divergence -- a lane of
// ec_stage pays every addition any lane of its wave needs, i.e. 255 doublings + ~255 additions; here the scalar is split with the
// curve endomorphism (k = k1 + k2 lambda, 128-bit halves, glv.cuh; phi(X, Y, ZZ, ZZZ) = (zeta X, Y, ZZ, ZZZ)) and walked as
// 129 doublings + the set bits of |k1| and |k2| (~128 additions).  The points of a wave are 2^(t+1) apart: 128-byte lines either way.
// UNIFORM = false: the last stages, one twiddle per lane -- the same split walk (130 doublings and, with 64 different scalars in
// a wave, practically every one of the 2 x 130 additions) against 255 + 255 for the plain walk.
template <int FB, int FS, bool UNIFORM>
__global__ void __launch_bounds__(256) ec_stage_glv(u32 *__restrict__ p, const u32 *__restrict__ tw, u32 half_n, int L, int t) {
    u32 low, hi;
    if (UNIFORM) {
        const u32 per_low = 1u << (L - 1 - t - 6);                   // waves per twiddle
        const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
        low = wave / per_low;
        hi = (wave % per_low) * 64 + lane;
    } else {
        const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
        if (q >= half_n) return;
        low = q & ((1u << t) - 1);
        hi = q >> t;
    }
    const u32 x0 = (hi << (t + 1)) | low, x1 = x0 + (1u << t);
    xyzz<FB> a = xyzz_load<FB>(p + 32 * (size_t)x0), b = xyzz_load<FB>(p + 32 * (size_t)x1);
    if (low != 0) {
        const fe w = fe_from_mont<FS>(fe_load(tw + 8 * ((size_t)low << (L - t - 1))));
        u32 m1[5], m2[5], n1, n2;
        glv_split<FS>(w, m1, n1, m2, n2);
        if (UNIFORM) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {                            // identical in every lane: say so (scalar registers, uniform branches)
                m1[i] = (u32)__builtin_amdgcn_readfirstlane((int)m1[i]);
                m2[i] = (u32)__builtin_amdgcn_readfirstlane((int)m2[i]);
            }
            n1 = (u32)__builtin_amdgcn_readfirstlane((int)n1);
            n2 = (u32)__builtin_amdgcn_readfirstlane((int)n2);
        }
        xyzz<FB> b1 = b, b2 = b;
        if (n1) b1.y = fe_neg<FB>(b.y);
        b2.x = fe_mulx<FB>(b.x, glv_zeta<FB>());
        if (n2) b2.y = fe_neg<FB>(b.y);
        xyzz<FB> r = xyzz_identity<FB>();
        for (int bit = 129; bit >= 0; --bit) {
            r = xyzz_dbl<FB>(r);
            const u32 w_ = (u32)bit >> 5, s_ = (u32)bit & 31;
            u32 l1 = 0, l2 = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                l1 = w_ == (u32)i ? m1[i] : l1;
                l2 = w_ == (u32)i ? m2[i] : l2;
            }
            if ((l1 >> s_) & 1) xyzz_add<FB>(r, b1);
            if ((l2 >> s_) & 1) xyzz_add<FB>(r, b2);
        }
        b = r;
    }
    xyzz<FB> s = a, d = a;
    xyzz_add<FB>(s, b);
    b.y = fe_neg<FB>(b.y);
    xyzz_add<FB>(d, b);
    xyzz_store<FB>(p + 32 * (size_t)x0, s);
    xyzz_store<FB>(p + 32 * (size_t)x1, d);
}

// out[i] = affine([minv] p[i]); minv is the same for every lane: host-computed NAF, divergence-free
template <int FB>
__global__ void __launch_bounds__(256) ec_scale_normalise(const u32 *__restrict__ p, u32 *__restrict__ out, u32 n,
                                                          const int8_t *__restrict__ naf, int top, int mont) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xyzz<FB> b = xyzz_load<FB>(p + 32 * (size_t)i), nb = b;
    nb.y = fe_neg<FB>(b.y);
    xyzz<FB> r = xyzz_identity<FB>();
    for (int k = top; k >= 0; --k) {
        r = xyzz_dbl<FB>(r);
        const int d = naf[k];
        if (d > 0) xyzz_add<FB>(r, b);
        else if (d < 0) xyzz_add<FB>(r, nb);
    }
    affine<FB> a = xyzz_to_affine<FB>(r);
    if (!mont) { a.x = fe_from_mont<FB>(a.x); a.y = fe_from_mont<FB>(a.y); }
    fe_store(out + 16 * (size_t)i, a.x);
    fe_store(out + 16 * (size_t)i + 8, a.y);
}

static int ec_naf(const u64 k_in[4], int8_t out[257]) {
    u64 k[5] = {k_in[0], k_in[1], k_in[2], k_in[3], 0};
    memset(out, 0, 257);
    int top = -1;
    for (int i = 0; i < 257; ++i) {
        if (k[0] & 1) {
            int d = 2 - (int)(k[0] & 3);
            out[i] = (int8_t)d;
            top = i;
            if (d > 0) k[0] -= 1;
            else for (int j = 0; j < 5; ++j) if (++k[j] != 0) break;
        }
        for (int j = 0; j < 4; ++j) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    return top;
}

template <int FB, int FS>
static int lagrange_basis_run(int field_s, const void *d_g, void *d_out, unsigned k, int form, hipStream_t st) {
    const u32 n = 1u << k;
    // alpha_inv = ROOT_OF_UNITY_INV^(2^(S-k)); ROOT_OF_UNITY = 5^((p-1)/2^32)
    const HostField &F = kHostField[field_s];
    u64 five[4] = {5, 0, 0, 0}, root[4], acc[4];
    host_mul(field_s, five, five, F.r2);
    memcpy(acc, F.one, 32);
    u64 base[4];
    memcpy(base, five, 32);
    {   // exponent (p - 1) >> 32 == p >> 32
        u64 e[4] = {(F.p[0] >> 32) | (F.p[1] << 32), (F.p[1] >> 32) | (F.p[2] << 32), (F.p[2] >> 32) | (F.p[3] << 32), F.p[3] >> 32};
        for (int i = 0; i < 256; i++) {
            if ((e[i / 64] >> (i % 64)) & 1) host_mul(field_s, acc, acc, base);
            host_mul(field_s, base, base, base);
        }
    }
    memcpy(root, acc, 32);
    for (unsigned i = k; i < 32; i++) host_mul(field_s, root, root, root);   // omega_k
    // inverse: omega_k^(2^k - 1)
    u64 inv[4];
    memcpy(inv, F.one, 32);
    {
        u64 b2[4];
        memcpy(b2, root, 32);
        for (unsigned i = 0; i < k; i++) {   // exponent 2^k - 1 = k ones
            host_mul(field_s, inv, inv, b2);
            host_mul(field_s, b2, b2, b2);
        }
    }
    // minv = 2^-k = TWO_INV^k; TWO_INV = (p + 1) / 2
    u64 two_inv[4] = {(F.p[0] >> 1) + 1, 0, 0, 0};
    {   // (p + 1) / 2 = (p >> 1) + 1 for odd p
        u64 h[4];
        for (int j = 0; j < 4; ++j) h[j] = (F.p[j] >> 1) | (j < 3 ? F.p[j + 1] << 63 : 0);
        u128 c = (u128)h[0] + 1;
        h[0] = (u64)c;
        for (int j = 1; j < 4 && (c >> 64); ++j) { c = (u128)h[j] + 1; h[j] = (u64)c; }
        host_mul(field_s, two_inv, h, F.r2);
    }
    u64 minv[4], minv_c[4];
    memcpy(minv, F.one, 32);
    for (unsigned i = 0; i < k; i++) host_mul(field_s, minv, minv, two_inv);
    host_from_mont(field_s, minv_c, minv);
    int8_t naf[257];
    int top = ec_naf(minv_c, naf);

    void *d_p = nullptr, *d_naf = nullptr;
    H2_HIP(hipMalloc(&d_p, (size_t)n * 128));
    H2_HIP(hipMalloc(&d_naf, 512));
    int rc = H2_OK;
    do {
        hipError_t e = hipMemcpyAsync(d_naf, naf, 257, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); rc = H2_ERR_HIP; break; }
        const u32 *d_tw = nullptr;
        if (k >= 1 && (rc = ntt_twiddle_table(field_s, (int)k, inv, st, &d_tw)) != H2_OK) break;
        dim3 block(256), gn((n + 255) / 256), gh((n / 2 + 255) / 256);
        hipLaunchKernelGGL((ec_load_bitrev<FB>), gn, block, 0, st, (const u32 *)d_g, (u32 *)d_p, n, (int)k, form == H2_FORM_MONTGOMERY);
        for (unsigned t = 0; t < k; ++t) {
            if (t + 7 <= k && k >= 9)      // a twiddle serves >= 64 butterflies: one twiddle per wave (n / 2 is a multiple of 256)
                hipLaunchKernelGGL((ec_stage_glv<FB, FS, true>), dim3(n / 2 / 256), block, 0, st, (u32 *)d_p, d_tw, n / 2, (int)k, (int)t);
            else if (t >= 2)     // per-lane twiddles; stages 0 and 1 multiply by 1 and by a fourth root of unity only
                hipLaunchKernelGGL((ec_stage_glv<FB, FS, false>), gh, block, 0, st, (u32 *)d_p, d_tw, n / 2, (int)k, (int)t);
            else
                hipLaunchKernelGGL((ec_stage<FB, FS>), gh, block, 0, st, (u32 *)d_p, d_tw, n / 2, (int)k, (int)t);
        }
        hipLaunchKernelGGL((ec_scale_normalise<FB>), gn, block, 0, st, (const u32 *)d_p, (u32 *)d_out, n, (const int8_t *)d_naf, top,
                           form == H2_FORM_MONTGOMERY);
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); rc = H2_ERR_HIP; }
    } while (0);
    (void)hipFree(d_p);
    (void)hipFree(d_naf);
    return rc;
}

}  // namespace h2

using namespace h2;

extern "C" int h2_lagrange_basis_device(int curve, const void *d_g_xy, void *d_out_xy, unsigned k, int form, void *stream) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !d_g_xy || !d_out_xy ||
        k >= 32 || d_g_xy == d_out_xy)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (curve == H2_PALLAS) return lagrange_basis_run<FP, FQ>(H2_FQ, d_g_xy, d_out_xy, k, form, (hipStream_t)stream);
    return lagrange_basis_run<FQ, FP>(H2_FP, d_g_xy, d_out_xy, k, form, (hipStream_t)stream);
}

extern "C" int h2_lagrange_basis(int curve, const uint64_t *g_xy, uint64_t *out_xy, unsigned k, int form) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !g_xy || !out_xy || k >= 32)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    const size_t bytes = (size_t)64 << k;
    void *d_in = nullptr, *d_out = nullptr;
    H2_HIP(hipMalloc(&d_in, bytes));
    hipError_t e = hipMalloc(&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, g_xy, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = h2_lagrange_basis_device(curve, d_in, d_out, k, form, nullptr);
        if (rc == H2_OK) e = hipMemcpy(out_xy, d_out, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}
