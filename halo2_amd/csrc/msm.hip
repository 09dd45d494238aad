// Pippenger multi-scalar multiplication for Pallas / Vesta on gfx950.
//
// Replaces the body of `best_multiexp` (halo2_proofs/src/arithmetic.rs:143-180) and `Buckets::sum`
// (:74-93).  The reference runs one CPU task per c-bit window, each re-streaming all n (scalar, base)
// pairs; the result is a group element, so any window width / digit encoding gives the same answer
// (SURVEY.md appendix A.1 item 7).  This implementation is organised for the GPU instead:
//
//   recode      one thread per scalar: Montgomery -> canonical once (the reference redoes `to_repr`
//               per window, :77), then W signed c-bit digits -> u16 codes, window-major        [HBM]
//   count       per (window, chunk) workgroup: 2^(c-1)-bin histogram in LDS (128 KiB at c = 16;
//               global atomics measured 20x slower), written out as per-chunk prefix slices    [LDS]
//   scan        per-bucket totals + exclusive offsets over all W * 2^(c-1) buckets
//   scatter     same workgroups: offsets in LDS, LDS atomics hand out slots; point index | sign
//               lands in the bucket-sorted entry list                                           [LDS]
//   accumulate  one thread per bucket: gather 64-B affine bases (MALL-resident at k = 20), mixed
//               XYZZ additions in registers -- the hot kernel, ~90 % of the modular multiplies  [VALU]
//   reduce      running-sum fold (:86-92) restructured as 8-bucket segments + a 15-bit scalar
//               multiple per segment, then a tree sum per window
//   combine     Horner over windows (:169-178) on one lane; emits Jacobian or affine
//
// No MFMA anywhere: this is modular-integer arithmetic.  Bound by VALU integer-multiply issue, not
// HBM: algorithmic traffic is 96 B per (scalar, base) pair against ~1.9e2 modular multiplies.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "common.h"
#include "curve.cuh"

namespace h2 {

static constexpr int kMaxC = 16;
static constexpr u32 kZeroCode = 0xFFFFu;
static constexpr int kSeg = 8;  // buckets per reduce segment

struct MsmShape {
    size_t n;
    int c;        // window bits
    int W;        // windows
    u32 NB;       // buckets per window = 2^(c-1)
    u32 B;        // chunks
    u32 chunk;    // scalars per chunk
};

static MsmShape msm_shape(size_t n) {
    MsmShape s;
    s.n = n;
    double best = 1e300;
    int bc = 4;
    for (int c = 4; c <= kMaxC; ++c) {
        int W = 255 / c + 1;
        double cost = (double)W * ((double)n * 10.5 + (double)(1u << (c - 1)) * 40.0);
        if (cost < best) { best = cost; bc = c; }
    }
    s.c = bc;
    s.W = 255 / bc + 1;
    s.NB = 1u << (bc - 1);
    u32 B = (u32)((n + 65535) / 65536);
    if (B < 1) B = 1;
    s.B = B;
    s.chunk = (u32)((n + B - 1) / B);
    return s;
}

__device__ __forceinline__ u32 limb_at(const fe &s, int idx) {
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v = (idx == i) ? s.v[i] : v;
    return v;
}

// ---- recode: scalars -> signed window digits -------------------------------------------------
// code = 0xFFFF for digit 0, else (|d| - 1) | (d < 0 ? 0x8000 : 0);  digits[w * n + i]
template <int FS>
__global__ void __launch_bounds__(256) msm_recode(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 n, int c, int W, int mont) {
    // n counts the optional extra (blind) scalar, which is element n - 1 and lives in its own buffer
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe s = (extra_scalar && i == n - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_from_mont<FS>(s);
    u32 carry = 0;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
    for (int w = 0; w < W; ++w) {
        int bit = w * c, word = bit >> 5, sh = bit & 31;
        u64 two = (u64)limb_at(s, word) | ((u64)limb_at(s, word + 1) << 32);  // limb_at(.., 8) = 0
        u32 raw = ((u32)(two >> sh) & mask) + carry;
        u32 code;
        if (raw > half) {
            carry = 1;
            code = ((1u << c) - raw - 1) | 0x8000u;   // d = raw - 2^c < 0, |d| - 1
        } else {
            carry = 0;
            code = raw ? raw - 1 : kZeroCode;
        }
        digits[(size_t)w * n + i] = (uint16_t)code;
    }
}

// ---- count: LDS histogram per (window, chunk) --------------------------------------------------
__global__ void __launch_bounds__(1024) msm_count(const uint16_t *__restrict__ digits, u32 *__restrict__ hist, u32 n,
                                                  u32 chunk, u32 NB) {
    extern __shared__ __attribute__((aligned(16))) u32 h[];
    const u32 b = blockIdx.x, w = blockIdx.y, B = gridDim.x;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) h[j] = 0;
    __syncthreads();
    u32 lo = b * chunk, hi = min(n, lo + chunk);
    const uint16_t *d = digits + (size_t)w * n;
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) atomicAdd(&h[code & 0x7FFFu], 1u);
    }
    __syncthreads();
    u32 *dst = hist + ((size_t)w * B + b) * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) dst[j] = h[j];
}

// ---- scan a: per-bucket totals, chunk slices become exclusive prefixes -------------------------
__global__ void __launch_bounds__(256) msm_chunk_prefix(u32 *__restrict__ hist, u32 *__restrict__ counts, u32 NB,
                                                        u32 B, u32 total_buckets) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_buckets) return;
    u32 w = g / NB, j = g % NB;
    u32 run = 0;
    for (u32 b = 0; b < B; ++b) {
        size_t k = ((size_t)w * B + b) * NB + j;
        u32 t = hist[k];
        hist[k] = run;
        run += t;
    }
    counts[g] = run;
}

// ---- scan b: exclusive scan of bucket totals (single workgroup) --------------------------------
__global__ void __launch_bounds__(1024) msm_scan_counts(const u32 *__restrict__ counts, u32 *__restrict__ starts,
                                                        u32 total) {
    __shared__ u32 part[1024];
    const u32 t = threadIdx.x;
    u32 per = (total + 1023) / 1024;
    u32 lo = t * per, hi = min(total, lo + per);
    u32 s = 0;
    for (u32 i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        u32 v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - s;
    for (u32 i = lo; i < hi; ++i) {
        starts[i] = run;
        run += counts[i];
    }
}

// ---- scatter: bucket-sorted entry list ----------------------------------------------------------
__global__ void __launch_bounds__(1024) msm_scatter(const uint16_t *__restrict__ digits, const u32 *__restrict__ hist,
                                                    const u32 *__restrict__ starts, u32 *__restrict__ entries, u32 n,
                                                    u32 chunk, u32 NB) {
    extern __shared__ __attribute__((aligned(16))) u32 off[];
    const u32 b = blockIdx.x, w = blockIdx.y, B = gridDim.x;
    const u32 *src = hist + ((size_t)w * B + b) * NB;
    const u32 *st = starts + (size_t)w * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) off[j] = st[j] + src[j];
    __syncthreads();
    u32 lo = b * chunk, hi = min(n, lo + chunk);
    const uint16_t *d = digits + (size_t)w * n;
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) {
            u32 pos = atomicAdd(&off[code & 0x7FFFu], 1u);
            entries[pos] = i | ((code & 0x8000u) << 16);
        }
    }
}

// ---- accumulate: one thread per bucket -----------------------------------------------------------
template <int FB>
__global__ void __launch_bounds__(256) msm_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, const u32 *__restrict__ counts,
                                                      u32 *__restrict__ buckets, u32 total_buckets) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_buckets) return;
    u32 s = starts[g], cnt = counts[g];
    xyzz<FB> acc = xyzz_identity<FB>();
    for (u32 k = 0; k < cnt; ++k) {
        u32 e = entries[s + k];
        u32 idx = e & 0x7FFFFFFFu;
        // the blind's base `w` (Params::commit, poly/commitment.rs:127) is element n, in its own buffer
        affine<FB> p = aff_load<FB>(idx == extra_index ? extra_base : bases + 16 * (size_t)idx);
        if (e >> 31) p.y = fe_neg<FB>(p.y);
        xyzz_madd<FB>(acc, p);
    }
    xyzz_store<FB>(buckets + 32 * (size_t)g, acc);
}

// k * p for a small k (bucket index offsets, < 2^16): MSB-first double-and-add
template <int FB> __device__ xyzz<FB> xyzz_mul_small(const xyzz<FB> &p, u32 k) {
    xyzz<FB> r = xyzz_identity<FB>();
    for (int b = 31 - __clz(k | 1); b >= 0; --b) {
        r = xyzz_dbl<FB>(r);
        if ((k >> b) & 1) xyzz_add<FB>(r, p);
    }
    return k ? r : xyzz_identity<FB>();
}

// ---- reduce level 1: segment of kSeg buckets -> sum_j (j+1) * B_j restricted to the segment ------
template <int FB>
__global__ void __launch_bounds__(256) msm_reduce_segments(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_segments) return;
    u32 segs_per_window = NB / kSeg;
    u32 w = t / segs_per_window, sg = t % segs_per_window;
    const u32 *base = buckets + 32 * ((size_t)w * NB + (size_t)sg * kSeg);
    xyzz<FB> run = xyzz_identity<FB>(), acc = xyzz_identity<FB>();
    for (int j = kSeg - 1; j >= 0; --j) {
        xyzz<FB> bk = xyzz_load<FB>(base + 32 * j);
        xyzz_add<FB>(run, bk);
        xyzz_add<FB>(acc, run);
    }
    // buckets of this segment carry weights sg*kSeg + (j+1): add (sg*kSeg) * run
    xyzz<FB> sh = xyzz_mul_small<FB>(run, sg * kSeg);
    xyzz_add<FB>(acc, sh);
    xyzz_store<FB>(partial + 32 * (size_t)t, acc);
}

// ---- reduce level 2: tree sum of a window's partials ----------------------------------------------
template <int FB>
__global__ void __launch_bounds__(256) msm_sum_window(const u32 *__restrict__ partial, u32 *__restrict__ window_sums,
                                                      u32 per_window) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 w = blockIdx.x, t = threadIdx.x;
    const u32 *src = partial + 32 * (size_t)w * per_window;
    xyzz<FB> acc = xyzz_identity<FB>();
    for (u32 i = t; i < per_window; i += blockDim.x) {
        xyzz<FB> p = xyzz_load<FB>(src + 32 * (size_t)i);
        xyzz_add<FB>(acc, p);
    }
    xyzz_store<FB>(sh + 32 * t, acc);
    __syncthreads();
    for (u32 off = blockDim.x / 2; off > 0; off >>= 1) {
        if (t < off) {
            xyzz<FB> a = xyzz_load<FB>(sh + 32 * t), b = xyzz_load<FB>(sh + 32 * (t + off));
            xyzz_add<FB>(a, b);
            xyzz_store<FB>(sh + 32 * t, a);
        }
        __syncthreads();
    }
    if (t == 0) {
        xyzz<FB> r = xyzz_load<FB>(sh);
        xyzz_store<FB>(window_sums + 32 * (size_t)w, r);
    }
}

// ---- combine: Horner over windows (+ optional blind term), emit Jacobian / affine -----------------
// extra: optional XYZZ point added at the end (the blind*w term of Params::commit)
template <int FB>
__global__ void msm_combine(const u32 *__restrict__ window_sums, int W, int c, const u32 *__restrict__ extra,
                            u32 *__restrict__ out, int out_kind, int out_mont) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    xyzz<FB> r = xyzz_identity<FB>();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) r = xyzz_dbl<FB>(r);
        xyzz<FB> s = xyzz_load<FB>(window_sums + 32 * (size_t)w);
        xyzz_add<FB>(r, s);
    }
    if (extra) {
        xyzz<FB> e = xyzz_load<FB>(extra);
        xyzz_add<FB>(r, e);
    }
    if (out_kind == H2_OUT_AFFINE) {
        affine<FB> a = xyzz_to_affine<FB>(r);
        if (!out_mont) { a.x = fe_from_mont<FB>(a.x); a.y = fe_from_mont<FB>(a.y); }
        fe_store(out, a.x);
        fe_store(out + 8, a.y);
    } else {
        fe X, Y, Z;
        xyzz_to_jacobian<FB>(r, X, Y, Z);
        if (!out_mont) { X = fe_from_mont<FB>(X); Y = fe_from_mont<FB>(Y); Z = fe_from_mont<FB>(Z); }
        fe_store(out, X);
        fe_store(out + 8, Y);
        fe_store(out + 16, Z);
    }
}

// ---- small helpers -----------------------------------------------------------------------------
// canonical -> Montgomery for n field elements / affine coordinates (in place)
template <int F> __global__ void __launch_bounds__(256) k_to_mont(u32 *a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store(a + 8 * i, fe_to_mont<F>(fe_load(a + 8 * i)));
}

// sum of Jacobian points (host helper for the multi-GPU partial sum): Jacobian -> XYZZ is
// (X, Y, Z^2, Z^3)
template <int FB>
__global__ void k_points_sum(const u32 *__restrict__ pts, u32 count, u32 *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    xyzz<FB> r = xyzz_identity<FB>();
    for (u32 i = 0; i < count; ++i) {
        const u32 *p = pts + 24 * (size_t)i;
        fe Z = fe_load(p + 16);
        if (fe_is_zero(Z)) continue;
        xyzz<FB> q;
        q.x = fe_load(p);
        q.y = fe_load(p + 8);
        q.zz = fe_sqr<FB>(Z);
        q.zzz = fe_mulx<FB>(q.zz, Z);
        xyzz_add<FB>(r, q);
    }
    fe X, Y, Z;
    xyzz_to_jacobian<FB>(r, X, Y, Z);
    fe_store(out, X);
    fe_store(out + 8, Y);
    fe_store(out + 16, Z);
}

// ---- host orchestration ----------------------------------------------------------------------------
struct MsmContext {
    std::mutex mu;
    DevBuf digits, hist, counts, starts, entries, buckets, partial, wsums, extra, stage_s, stage_b, out, small;
    bool attr_set = false;
};

// One workspace per (device, stream): calls enqueued on different streams never share scratch.
static MsmContext &msm_ctx(hipStream_t st = nullptr) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::unique_ptr<MsmContext>> ctxs;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto &slot = ctxs[std::make_pair(dev, st)];
    if (!slot) slot.reset(new MsmContext());
    return *slot;
}

template <int FB, int FS>
static int msm_launch(MsmContext &cx, const void *d_scalars, const void *d_bases, size_t n_in, int form,
                      const void *d_extra_scalar, const void *d_extra_base, int out_kind, void *d_out, hipStream_t st) {
    const u32 *d_extra = nullptr;
    const size_t n = n_in + (d_extra_scalar ? 1 : 0);
    if (n == 0) {
        // identity (+ extra)
        int rc;
        if ((rc = cx.wsums.reserve(128)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.wsums.ptr, 0, 128, st));
        hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, cx.wsums.as<u32>(), 1, 1, d_extra, (u32 *)d_out,
                           out_kind, form == H2_FORM_MONTGOMERY);
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    MsmShape sh = msm_shape(n);
    const u32 total_buckets = (u32)sh.W * sh.NB;
    const u32 segs = total_buckets / kSeg;
    int rc;
    if ((rc = cx.digits.reserve((size_t)sh.W * n * 2)) != H2_OK) return rc;
    if ((rc = cx.hist.reserve((size_t)sh.W * sh.B * sh.NB * 4)) != H2_OK) return rc;
    if ((rc = cx.counts.reserve((size_t)total_buckets * 4)) != H2_OK) return rc;
    if ((rc = cx.starts.reserve((size_t)total_buckets * 4)) != H2_OK) return rc;
    if ((rc = cx.entries.reserve((size_t)sh.W * n * 4)) != H2_OK) return rc;
    if ((rc = cx.buckets.reserve((size_t)total_buckets * 128)) != H2_OK) return rc;
    if ((rc = cx.partial.reserve((size_t)segs * 128)) != H2_OK) return rc;
    if ((rc = cx.wsums.reserve((size_t)sh.W * 128)) != H2_OK) return rc;
    if (!cx.attr_set) {
        H2_HIP(hipFuncSetAttribute((const void *)msm_count, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        H2_HIP(hipFuncSetAttribute((const void *)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        cx.attr_set = true;
    }
    const u32 n32 = (u32)n;
    prof_begin(PROF_MSM_SORT, st);
    hipLaunchKernelGGL((msm_recode<FS>), dim3((n32 + 255) / 256), dim3(256), 0, st, (const u32 *)d_scalars,
                       (const u32 *)d_extra_scalar, cx.digits.as<uint16_t>(), n32, sh.c, sh.W,
                       form == H2_FORM_MONTGOMERY);
    hipLaunchKernelGGL(msm_count, dim3(sh.B, sh.W), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(),
                       cx.hist.as<u32>(), n32, sh.chunk, sh.NB);
    hipLaunchKernelGGL(msm_chunk_prefix, dim3((total_buckets + 255) / 256), dim3(256), 0, st, cx.hist.as<u32>(),
                       cx.counts.as<u32>(), sh.NB, sh.B, total_buckets);
    hipLaunchKernelGGL(msm_scan_counts, dim3(1), dim3(1024), 0, st, cx.counts.as<u32>(), cx.starts.as<u32>(),
                       total_buckets);
    hipLaunchKernelGGL(msm_scatter, dim3(sh.B, sh.W), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(),
                       cx.hist.as<u32>(), cx.starts.as<u32>(), cx.entries.as<u32>(), n32, sh.chunk, sh.NB);
    prof_end(PROF_MSM_SORT, st);
    prof_begin(PROF_MSM_ACCUMULATE, st);
    hipLaunchKernelGGL((msm_accumulate<FB>), dim3((total_buckets + 255) / 256), dim3(256), 0, st, (const u32 *)d_bases,
                       (const u32 *)d_extra_base, d_extra_base ? (u32)n_in : 0xFFFFFFFFu, cx.entries.as<u32>(),
                       cx.starts.as<u32>(), cx.counts.as<u32>(), cx.buckets.as<u32>(), total_buckets);
    prof_end(PROF_MSM_ACCUMULATE, st);
    prof_begin(PROF_MSM_REDUCE, st);
    hipLaunchKernelGGL((msm_reduce_segments<FB>), dim3((segs + 255) / 256), dim3(256), 0, st, cx.buckets.as<u32>(),
                       cx.partial.as<u32>(), sh.NB, segs);
    hipLaunchKernelGGL((msm_sum_window<FB>), dim3(sh.W), dim3(256), 256 * 128, st, cx.partial.as<u32>(),
                       cx.wsums.as<u32>(), sh.NB / kSeg);
    hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, cx.wsums.as<u32>(), sh.W, sh.c, d_extra,
                       (u32 *)d_out, out_kind, form == H2_FORM_MONTGOMERY);
    prof_end(PROF_MSM_REDUCE, st);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

static int msm_dispatch(MsmContext &cx, int curve, const void *d_scalars, const void *d_bases, size_t n, int form,
                        const void *d_extra_scalar, const void *d_extra_base, int out_kind, void *d_out, hipStream_t st) {
    if (curve == H2_PALLAS)
        return msm_launch<FP, FQ>(cx, d_scalars, d_bases, n, form, d_extra_scalar, d_extra_base, out_kind, d_out, st);
    return msm_launch<FQ, FP>(cx, d_scalars, d_bases, n, form, d_extra_scalar, d_extra_base, out_kind, d_out, st);
}

// ---- registered bases -------------------------------------------------------------------------------
struct Bases {
    int curve;
    size_t n;
    void *d_xy;  // Montgomery affine, n * 64 B
};
static std::mutex g_bases_mu;
static std::map<h2_bases_t, std::shared_ptr<Bases>> g_bases;
static h2_bases_t g_next_handle = 1;

static std::shared_ptr<Bases> find_bases(h2_bases_t h) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(h);
    return it == g_bases.end() ? nullptr : it->second;
}

static bool bad_common(int curve, int form, int out_kind) {
    return (curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) ||
           (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE);
}

// device-side commit core: optional (blind, w) pair rides along as element n; bases Montgomery on device
static int commit_core(MsmContext &cx, int curve, const void *d_scalars, const void *d_bases, size_t n, int form,
                       const void *d_w, const void *d_blind, int out_kind, void *d_out, hipStream_t st) {
    return msm_dispatch(cx, curve, d_scalars, d_bases, n, form, d_blind, d_w, out_kind, d_out, st);
}

}  // namespace h2

using namespace h2;

extern "C" int h2_msm_window_bits(size_t n) { return msm_shape(n ? n : 1).c; }

extern "C" int h2_msm_device(int curve, const void *d_scalars, const void *d_bases_xy, size_t n, int form, int out_kind,
                             void *d_out, void *stream) {
    if (bad_common(curve, form, out_kind) || !d_out || (n && (!d_scalars || !d_bases_xy)) || n > 0x7FFFFFFFu)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const void *bases = d_bases_xy;
    if (form == H2_FORM_CANONICAL && n) {
        // bases arrive canonical: convert a private copy to Montgomery
        if ((rc = cx.stage_b.reserve(n * 64)) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(cx.stage_b.ptr, d_bases_xy, n * 64, hipMemcpyDeviceToDevice, st));
        size_t cnt = n * 2;
        if (curve == H2_PALLAS)
            hipLaunchKernelGGL((k_to_mont<FP>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cx.stage_b.as<u32>(), cnt);
        else
            hipLaunchKernelGGL((k_to_mont<FQ>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cx.stage_b.as<u32>(), cnt);
        bases = cx.stage_b.ptr;
    }
    return msm_dispatch(cx, curve, d_scalars, bases, n, form, nullptr, nullptr, out_kind, d_out, st);
}

extern "C" int h2_msm(int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, int form, int out_kind,
                      uint64_t *out) {
    if (bad_common(curve, form, out_kind) || !out || (n && (!scalars || !bases_xy)) || n > 0x7FFFFFFFu) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    MsmContext &cx = msm_ctx();
    const size_t out_bytes = out_kind == H2_OUT_AFFINE ? 64 : 96;
    {
        std::lock_guard<std::mutex> lk(cx.mu);
        if ((rc = cx.stage_s.reserve(n * 32 + 32)) != H2_OK) return rc;
        if ((rc = cx.stage_b.reserve(n * 64 + 64)) != H2_OK) return rc;
        if ((rc = cx.out.reserve(128)) != H2_OK) return rc;
        if (n) {
            H2_HIP(hipMemcpyAsync(cx.stage_s.ptr, scalars, n * 32, hipMemcpyHostToDevice, 0));
            H2_HIP(hipMemcpyAsync(cx.stage_b.ptr, bases_xy, n * 64, hipMemcpyHostToDevice, 0));
            if (form == H2_FORM_CANONICAL) {
                size_t cnt = n * 2;
                if (curve == H2_PALLAS)
                    hipLaunchKernelGGL((k_to_mont<FP>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, cx.stage_b.as<u32>(), cnt);
                else
                    hipLaunchKernelGGL((k_to_mont<FQ>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, cx.stage_b.as<u32>(), cnt);
            }
        }
        rc = msm_dispatch(cx, curve, cx.stage_s.ptr, cx.stage_b.ptr, n, form, nullptr, nullptr, out_kind, cx.out.ptr, 0);
        if (rc != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(out, cx.out.ptr, out_bytes, hipMemcpyDeviceToHost, 0));
        H2_HIP(hipStreamSynchronize(0));
    }
    return H2_OK;
}

extern "C" int h2_bases_register(int curve, const uint64_t *bases_xy, size_t n, int form, h2_bases_t *handle) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) ||
        !handle || (n && !bases_xy) || n > 0x7FFFFFFEu)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    auto b = std::make_shared<Bases>();
    b->curve = curve;
    b->n = n;
    b->d_xy = nullptr;
    H2_HIP(hipMalloc(&b->d_xy, n * 64 + 64));
    if (n) {
        H2_HIP(hipMemcpy(b->d_xy, bases_xy, n * 64, hipMemcpyHostToDevice));
        if (form == H2_FORM_CANONICAL) {
            size_t cnt = n * 2;
            if (curve == H2_PALLAS)
                hipLaunchKernelGGL((k_to_mont<FP>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, (u32 *)b->d_xy, cnt);
            else
                hipLaunchKernelGGL((k_to_mont<FQ>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, (u32 *)b->d_xy, cnt);
            H2_HIP(hipStreamSynchronize(0));
        }
    }
    std::lock_guard<std::mutex> lk(g_bases_mu);
    h2_bases_t h = g_next_handle++;
    g_bases[h] = b;
    *handle = h;
    return H2_OK;
}

extern "C" int h2_bases_free(h2_bases_t handle) {
    std::shared_ptr<Bases> b;
    {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        auto it = g_bases.find(handle);
        if (it == g_bases.end()) return H2_ERR_HANDLE;
        b = it->second;
        g_bases.erase(it);
    }
    if (b->d_xy) H2_HIP(hipFree(b->d_xy));
    return H2_OK;
}

extern "C" int h2_commit_device(h2_bases_t g, const void *d_scalars, size_t n, const void *d_w_xy, const void *d_blind,
                                int form, int out_kind, void *d_out, void *stream) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !d_out || (n && !d_scalars) || n > b->n || ((d_w_xy == nullptr) != (d_blind == nullptr)))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const void *w = d_w_xy;
    if (d_w_xy && form == H2_FORM_CANONICAL) {
        if ((rc = cx.small.reserve(64)) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(cx.small.ptr, d_w_xy, 64, hipMemcpyDeviceToDevice, st));
        if (b->curve == H2_PALLAS) hipLaunchKernelGGL((k_to_mont<FP>), dim3(1), dim3(256), 0, st, cx.small.as<u32>(), (size_t)2);
        else hipLaunchKernelGGL((k_to_mont<FQ>), dim3(1), dim3(256), 0, st, cx.small.as<u32>(), (size_t)2);
        w = cx.small.ptr;
    }
    return commit_core(cx, b->curve, d_scalars, b->d_xy, n, form, w, d_blind, out_kind, d_out, st);
}

extern "C" int h2_commit(h2_bases_t g, const uint64_t *scalars, size_t n, const uint64_t *w_xy, const uint64_t *blind,
                         int form, int out_kind, uint64_t *out) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !out || (n && !scalars) || n > b->n || ((w_xy == nullptr) != (blind == nullptr)))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    MsmContext &cx = msm_ctx();
    const size_t out_bytes = out_kind == H2_OUT_AFFINE ? 64 : 96;
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.stage_s.reserve(n * 32 + 32)) != H2_OK) return rc;
    if ((rc = cx.small.reserve(64 + 32)) != H2_OK) return rc;
    if ((rc = cx.out.reserve(128)) != H2_OK) return rc;
    if (n) H2_HIP(hipMemcpyAsync(cx.stage_s.ptr, scalars, n * 32, hipMemcpyHostToDevice, 0));
    const void *d_w = nullptr, *d_bl = nullptr;
    if (w_xy) {
        H2_HIP(hipMemcpyAsync(cx.small.ptr, w_xy, 64, hipMemcpyHostToDevice, 0));
        H2_HIP(hipMemcpyAsync((char *)cx.small.ptr + 64, blind, 32, hipMemcpyHostToDevice, 0));
        if (form == H2_FORM_CANONICAL) {
            if (b->curve == H2_PALLAS) hipLaunchKernelGGL((k_to_mont<FP>), dim3(1), dim3(256), 0, 0, cx.small.as<u32>(), (size_t)2);
            else hipLaunchKernelGGL((k_to_mont<FQ>), dim3(1), dim3(256), 0, 0, cx.small.as<u32>(), (size_t)2);
        }
        d_w = cx.small.ptr;
        d_bl = (char *)cx.small.ptr + 64;
    }
    rc = commit_core(cx, b->curve, cx.stage_s.ptr, b->d_xy, n, form, d_w, d_bl, out_kind, cx.out.ptr, 0);
    if (rc != H2_OK) return rc;
    H2_HIP(hipMemcpyAsync(out, cx.out.ptr, out_bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

extern "C" int h2_points_sum(int curve, const uint64_t *points_xyz, size_t count, uint64_t *out_xyz) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || !out_xyz || (count && !points_xyz) || count > (1u << 20)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    MsmContext &cx = msm_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.stage_b.reserve(count * 96 + 96)) != H2_OK) return rc;
    if ((rc = cx.out.reserve(128)) != H2_OK) return rc;
    if (count) H2_HIP(hipMemcpyAsync(cx.stage_b.ptr, points_xyz, count * 96, hipMemcpyHostToDevice, 0));
    if (curve == H2_PALLAS) hipLaunchKernelGGL((k_points_sum<FP>), dim3(1), dim3(64), 0, 0, cx.stage_b.as<u32>(), (u32)count, cx.out.as<u32>());
    else hipLaunchKernelGGL((k_points_sum<FQ>), dim3(1), dim3(64), 0, 0, cx.stage_b.as<u32>(), (u32)count, cx.out.as<u32>());
    H2_HIP(hipMemcpyAsync(out_xyz, cx.out.ptr, 96, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}
