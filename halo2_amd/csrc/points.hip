// Compressed point encoding of the URS file (`Params::write` / `Params::read`, halo2_proofs/src/poly/commitment.rs:169-205)
// and of every point a proof carries (`write_point`, transcript.rs:183-187): pasta_curves' `to_bytes` / `from_bytes` --
// x as 32 little-endian bytes with the parity of y in the top bit; the identity is 32 zero bytes.
//
//   h2_points_compress     n affine points -> n x 32 bytes           (2 Montgomery reductions per point)
//   h2_points_decompress   n x 32 bytes -> n affine points           (one square root in the base field per point)
//
// Reading a k = 20 URS means 2^21 square roots; p - 1 = 2^32 T for both Pasta fields, so the root is Tonelli-Shanks:
// a^((T-1)/2) (222 squarings), then at most 32 correction rounds.  One lane per point, 32-byte / 64-byte contiguous
// accesses per lane, no shared state.  ~10 ms for the whole file on one MI355X against seconds on the host.
#include <vector>

#include "common.h"
#include "field_sqrt.cuh"

namespace h2 {

template <int FB> __global__ void __launch_bounds__(256) points_compress(const u32 *__restrict__ xy, size_t n, int mont, u32 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(xy + 16 * i), y = fe_load(xy + 16 * i + 8);
    if (mont) {
        x = fe_from_mont<FB>(x);
        y = fe_from_mont<FB>(y);
    }
    x.v[7] |= (y.v[0] & 1u) << 31;         // identity (0, 0) stays all zero
    fe_store(out + 8 * i, x);
}

template <int FB> __global__ void __launch_bounds__(256) points_decompress(const u32 *__restrict__ in, size_t n, int mont, u32 *__restrict__ xy,
                                                                            u32 *__restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(in + 8 * i);
    const u32 sign = x.v[7] >> 31;
    x.v[7] &= 0x7fffffffu;
    fe xo = fe_zero(), yo = fe_zero();
    bool ok = !fe_eq(fe_reduce_once<FB>(x), x) ? false : true;     // x must already be < p
    if (ok && fe_is_zero(x)) {
        ok = sign == 0;                                             // the identity; (0, odd) is not a point
    } else if (ok) {
        const fe xm = fe_to_mont<FB>(x);
        const fe rhs = fe_add<FB>(fe_mulx<FB>(fe_sqr<FB>(xm), xm), curve_b<FB>());
        fe y;
        ok = fe_sqrt<FB>(rhs, y);
        if (ok) {
            fe yc = fe_from_mont<FB>(y);
            if ((yc.v[0] & 1u) != sign) {
                y = fe_neg<FB>(y);
                yc = fe_from_mont<FB>(y);
            }
            xo = mont ? xm : x;
            yo = mont ? y : yc;
        }
    }
    if (!ok) atomicOr(bad, 1u);
    fe_store(xy + 16 * i, xo);
    fe_store(xy + 16 * i + 8, yo);
}

namespace {
bool bad_curve_form(int curve, int form) {
    return (curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY);
}
constexpr size_t kMaxPoints = (size_t)1 << 30;

int compress_launch(int curve, const void *d_xy, size_t n, int form, void *d_out, hipStream_t st) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int mont = form == H2_FORM_MONTGOMERY;
    if (curve == H2_PALLAS) hipLaunchKernelGGL((points_compress<FP>), grid, block, 0, st, (const u32 *)d_xy, n, mont, (u32 *)d_out);
    else hipLaunchKernelGGL((points_compress<FQ>), grid, block, 0, st, (const u32 *)d_xy, n, mont, (u32 *)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// synchronises `st`: the verdict on the encodings has to reach the host
int decompress_launch(int curve, const void *d_in, size_t n, int form, void *d_xy, hipStream_t st) {
    u32 *d_bad = nullptr;
    H2_HIP(hipMalloc(&d_bad, 4));
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, st);
    u32 bad = 0;
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        const int mont = form == H2_FORM_MONTGOMERY;
        if (curve == H2_PALLAS) hipLaunchKernelGGL((points_decompress<FP>), grid, block, 0, st, (const u32 *)d_in, n, mont, (u32 *)d_xy, d_bad);
        else hipLaunchKernelGGL((points_decompress<FQ>), grid, block, 0, st, (const u32 *)d_in, n, mont, (u32 *)d_xy, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_bad);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return bad ? H2_ERR_DECODE : H2_OK;
}
}  // namespace
}  // namespace h2

using namespace h2;

extern "C" int h2_points_compress_device(int curve, const void *d_xy, size_t n, int form, void *d_out, void *stream) {
    if (bad_curve_form(curve, form) || (n && (!d_xy || !d_out)) || n > kMaxPoints) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    return compress_launch(curve, d_xy, n, form, d_out, (hipStream_t)stream);
}

extern "C" int h2_points_compress(int curve, const uint64_t *xy, size_t n, int form, uint8_t *out) {
    if (bad_curve_form(curve, form) || (n && (!xy || !out)) || n > kMaxPoints) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    void *d_in = nullptr, *d_out = nullptr;
    H2_HIP(hipMalloc(&d_in, n * 64));
    hipError_t e = hipMalloc(&d_out, n * 32);
    if (e == hipSuccess) e = hipMemcpy(d_in, xy, n * 64, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = compress_launch(curve, d_in, n, form, d_out, 0);
        if (rc == H2_OK) e = hipMemcpy(out, d_out, n * 32, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}

extern "C" int h2_points_decompress_device(int curve, const void *d_bytes, size_t n, int form, void *d_out_xy, void *stream) {
    if (bad_curve_form(curve, form) || (n && (!d_bytes || !d_out_xy)) || n > kMaxPoints) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    return decompress_launch(curve, d_bytes, n, form, d_out_xy, (hipStream_t)stream);
}

extern "C" int h2_points_decompress(int curve, const uint8_t *bytes, size_t n, int form, uint64_t *out_xy) {
    if (bad_curve_form(curve, form) || (n && (!bytes || !out_xy)) || n > kMaxPoints) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    void *d_in = nullptr, *d_out = nullptr;
    H2_HIP(hipMalloc(&d_in, n * 32));
    hipError_t e = hipMalloc(&d_out, n * 64);
    if (e == hipSuccess) e = hipMemcpy(d_in, bytes, n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = decompress_launch(curve, d_in, n, form, d_out, 0);
        if (rc == H2_OK) e = hipMemcpy(out_xy, d_out, n * 64, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    return rc;
}
