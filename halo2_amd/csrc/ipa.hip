// Inner-product-argument round kernels: the per-round work of
// `commitment::create_proof` (halo2_proofs/src/poly/commitment/prover.rs:100-142) that is not an MSM.
//
//   h2_generator_collapse   parallel_generator_collapse (:154-166):  g'[i] = g_lo[i] + [u_j] * g_hi[i], normalised to
//                           affine -- n / 2^(j+1) FULL 255-bit scalar multiplications per round, an order of
//                           magnitude more group operations per proof than one commit.  One lane per point; the
//                           challenge is the same for every lane, so its NAF recoding (done once on the host, ~85
//                           non-zero digits) drives a divergence-free double-and-add; one Fermat inversion per lane
//                           normalises (10 % of the lane's work).
//   h2_fold_scalars         the `p'` / `b` collapse (:128-131):  a[i] += a[i + half] * factor.
//
// Both keep their vectors on the device across rounds (d_* variants), removing 2k host round trips per proof.
#include <vector>

#include "common.h"
#include "curve.cuh"
#include "host_field.h"

namespace h2 {

// naf: 256 signed digits in {-1, 0, 1}, little-endian, uniform across lanes
template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse(u32 *__restrict__ g, u32 half, const int8_t *__restrict__ naf, int top) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const affine<FB> hi = aff_load<FB>(g + 16 * (size_t)(half + i));
    affine<FB> hi_neg = hi;
    hi_neg.y = fe_neg<FB>(hi.y);
    xyzz<FB> acc = xyzz_identity<FB>();
    for (int b = top; b >= 0; --b) {           // uniform control flow: every lane walks the same digits
        acc = xyzz_dbl<FB>(acc);
        const int d = naf[b];
        if (d > 0) xyzz_madd<FB>(acc, hi);
        else if (d < 0) xyzz_madd<FB>(acc, hi_neg);
    }
    const affine<FB> lo = aff_load<FB>(g + 16 * (size_t)i);
    xyzz_madd<FB>(acc, lo);
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

struct IpaContext {
    std::mutex mu;
    DevBuf naf, stage;
};
static IpaContext &ipa_ctx() {
    static IpaContext c;
    return c;
}

// non-adjacent form of a canonical 256-bit scalar; returns the index of the top non-zero digit (-1 for zero)
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

static int collapse_launch(int curve, void *d_g, size_t half, const u64 *u, int form, hipStream_t st) {
    IpaContext &cx = ipa_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    const int sf = curve == H2_PALLAS ? H2_FQ : H2_FP;      // the challenge lives in the scalar field
    u64 canon[4];
    if (form == H2_FORM_MONTGOMERY) host_from_mont(sf, canon, u);
    else memcpy(canon, u, 32);
    int8_t naf[257];
    int top = naf_recode(canon, naf);
    int rc = cx.naf.reserve(512);
    if (rc != H2_OK) return rc;
    // the digit buffer is reused across calls: order the copy after earlier kernels that read it
    H2_HIP(hipStreamSynchronize(st));
    H2_HIP(hipMemcpyAsync(cx.naf.ptr, naf, 257, hipMemcpyHostToDevice, st));
    dim3 grid((unsigned)((half + 255) / 256)), block(256);
    if (form == H2_FORM_CANONICAL) {
        dim3 g2((unsigned)((half * 4 + 255) / 256));
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_to_mont<FP>), g2, block, 0, st, (u32 *)d_g, half * 4, 1);
        else hipLaunchKernelGGL((ipa_to_mont<FQ>), g2, block, 0, st, (u32 *)d_g, half * 4, 1);
    }
    if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_collapse<FP>), grid, block, 0, st, (u32 *)d_g, (u32)half, cx.naf.as<int8_t>(), top);
    else hipLaunchKernelGGL((ipa_collapse<FQ>), grid, block, 0, st, (u32 *)d_g, (u32)half, cx.naf.as<int8_t>(), top);
    if (form == H2_FORM_CANONICAL) {
        dim3 g2((unsigned)((half * 2 + 255) / 256));
        if (curve == H2_PALLAS) hipLaunchKernelGGL((ipa_to_mont<FP>), g2, block, 0, st, (u32 *)d_g, half * 2, 0);
        else hipLaunchKernelGGL((ipa_to_mont<FQ>), g2, block, 0, st, (u32 *)d_g, half * 2, 0);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
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
