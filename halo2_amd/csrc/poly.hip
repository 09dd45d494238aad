// Polynomial helpers either side of the MSM / NTT path: the O(n) field loops the prover runs
// between its commits and transforms, so that a column can stay in HBM from witness to opening.
//
//   h2_eval_polynomial   arithmetic.rs:298-303   Horner evaluation
//   h2_inner_product     arithmetic.rs:308-318   <a, b>
//   h2_kate_division     arithmetic.rs:322-341   a(X) / (X - b), no remainder
//   h2_powers            poly/commitment/prover.rs:90-97   1, x, x^2, ...
//   h2_scale_add         poly/commitment/prover.rs:70      a <- a * x + b   (P' = S * xi + P)
//   h2_batch_invert      ff::BatchInvert as used at plonk/permutation/prover.rs:118, plonk/lookup/prover.rs:297
//   h2_grand_product     plonk/permutation/prover.rs:147-153   z[0] = init, z[i] = z[i-1] * m[i-1]
//
// All are HBM-bound (32 B in, 32 B out per element, tens of modular multiplications at most), so the layout rule is
// the only one that matters: every global access is a coalesced 16-byte-per-lane stream.  The three scans
// (division, grand product, batch inversion) need each lane to own CONSECUTIVE elements; a 2048-element tile is
// therefore staged through LDS (coalesced in, lane-private chunks of 8 out, odd row pitch = no bank conflicts).
// Field arithmetic is exact, so any association order gives the reference's result bit for bit.
#include <vector>

#include "common.h"
#include "field.cuh"
#include "host_field.h"

namespace h2 {

constexpr int kPT = 256;             // lanes per workgroup
constexpr int kPC = 8;               // consecutive elements per lane
constexpr int kTile = kPT * kPC;     // elements per tile
constexpr int kPitch = 8 * kPC + 1;  // LDS words per lane row
constexpr int kSPitch = 9;           // scan array pitch

__device__ __forceinline__ fe lds_get(const u32 *p) { return fe{{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]}}; }
__device__ __forceinline__ void lds_put(u32 *p, const fe &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = a.v[i];
}

// tile element j (global index base + j) <-> sh[(j / kPC) * kPitch + (j % kPC) * 8 ...]; indices >= n read `fill`
__device__ __forceinline__ void tile_load(u32 *sh, const u32 *g, size_t base, size_t n, u32 fill0) {
    for (int it = 0; it < 2 * kPC; ++it) {
        const u32 q = it * kPT + threadIdx.x, j = q >> 1, half = q & 1;
        const size_t gi = base + j;
        uint4 v = make_uint4(half ? 0 : fill0, 0, 0, 0);
        if (gi < n) v = reinterpret_cast<const uint4 *>(g)[gi * 2 + half];
        u32 *d = sh + (j / kPC) * kPitch + (j % kPC) * 8 + half * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
// stores tile positions whose global index i = base + j lies in [lo, hi) to g[i - shift]
__device__ __forceinline__ void tile_store(const u32 *sh, u32 *g, size_t base, size_t lo, size_t hi, size_t shift) {
    for (int it = 0; it < 2 * kPC; ++it) {
        const u32 q = it * kPT + threadIdx.x, j = q >> 1, half = q & 1;
        const size_t gi = base + j;
        if (gi < lo || gi >= hi) continue;
        const u32 *s = sh + (j / kPC) * kPitch + (j % kPC) * 8 + half * 4;
        reinterpret_cast<uint4 *>(g)[(gi - shift) * 2 + half] = make_uint4(s[0], s[1], s[2], s[3]);
    }
}

template <int F> __device__ fe fe_pow_u32(const fe &b, u32 e) {
    fe acc = fe_one<F>();
    if (!e) return acc;
    for (int i = 31 - __clz(e); i >= 0; --i) {
        acc = fe_sqr<F>(acc);
        if ((e >> i) & 1) acc = fe_mulx<F>(acc, b);
    }
    return acc;
}

// ---- reductions ----------------------------------------------------------------------------------------------------
template <int F> __device__ __forceinline__ void block_sum_to(u32 *sc, fe v, u32 *dst) {
    lds_put(sc + threadIdx.x * kSPitch, v);
    __syncthreads();
    for (int off = kPT / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            fe a = lds_get(sc + threadIdx.x * kSPitch), b = lds_get(sc + (threadIdx.x + off) * kSPitch);
            lds_put(sc + threadIdx.x * kSPitch, fe_add<F>(a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(dst, lds_get(sc));
}

// partial[blk] = sum over the lanes' interleaved elements of a[i] * b[i]  (raw Montgomery products)
template <int F> __global__ void __launch_bounds__(kPT) poly_inner_partial(const u32 *__restrict__ a, const u32 *__restrict__ b, size_t n,
                                                                           u32 *__restrict__ partial) {
    __shared__ u32 sc[kPT * kSPitch];
    const size_t T = (size_t)gridDim.x * kPT;
    fe acc = fe_zero();
    for (size_t i = (size_t)blockIdx.x * kPT + threadIdx.x; i < n; i += T)
        acc = fe_add<F>(acc, fe_mulx<F>(fe_load(a + 8 * i), fe_load(b + 8 * i)));
    block_sum_to<F>(sc, acc, partial + 8 * (size_t)blockIdx.x);
}

// lane t evaluates the sub-polynomial of the coefficients i = t (mod T) by Horner in x^T, then scales by x^t
template <int F> __global__ void __launch_bounds__(kPT) poly_eval_partial(const u32 *__restrict__ a, size_t n, fe x, fe xT,
                                                                          u32 *__restrict__ partial) {
    __shared__ u32 sc[kPT * kSPitch];
    const size_t T = (size_t)gridDim.x * kPT, t = (size_t)blockIdx.x * kPT + threadIdx.x;
    fe acc = fe_zero();
    if (t < n) {
        const size_t top = (n - 1 - t) / T;
        for (size_t j = top + 1; j-- > 0;) acc = fe_add<F>(fe_mulx<F>(acc, xT), fe_load(a + 8 * (t + j * T)));
        acc = fe_mulx<F>(acc, fe_pow_u32<F>(x, (u32)t));
    }
    block_sum_to<F>(sc, acc, partial + 8 * (size_t)blockIdx.x);
}

// out = post(sum of partials); post: 0 = as is, 1 = multiply by R^2 (raw product of two canonical inputs -> canonical)
template <int F> __global__ void __launch_bounds__(kPT) poly_sum_partials(const u32 *__restrict__ partial, u32 count, int post, u32 *__restrict__ out) {
    __shared__ u32 sc[kPT * kSPitch];
    fe acc = fe_zero();
    for (u32 i = threadIdx.x; i < count; i += kPT) acc = fe_add<F>(acc, fe_load(partial + 8 * (size_t)i));
    block_sum_to<F>(sc, acc, out);
    if (post == 1 && threadIdx.x == 0) {
        fe v = fe_load(out);
        fe_store(out, fe_mulx<F>(v, fe_r2<F>()));   // x R^2 twice over R: the raw sum is (a.b) / R
    }
}

// ---- elementwise ---------------------------------------------------------------------------------------------------
template <int F> __global__ void __launch_bounds__(kPT) poly_scale_add(u32 *__restrict__ a, const u32 *__restrict__ b, size_t n, fe x) {
    const size_t i = (size_t)blockIdx.x * kPT + threadIdx.x;
    if (i >= n) return;
    fe_store(a + 8 * i, fe_add<F>(fe_mulx<F>(fe_load(a + 8 * i), x), fe_load(b + 8 * i)));
}

template <int F> __global__ void __launch_bounds__(kPT) poly_powers(u32 *__restrict__ out, size_t n, fe x, int canonical) {
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const size_t base = (size_t)blockIdx.x * kTile;
    fe cur = fe_pow_u32<F>(x, (u32)(base + (size_t)threadIdx.x * kPC));
    u32 *row = sh + threadIdx.x * kPitch;
#pragma unroll 1
    for (int e = 0; e < kPC; ++e) {
        lds_put(row + 8 * e, canonical ? fe_from_mont<F>(cur) : cur);
        cur = fe_mulx<F>(cur, x);
    }
    __syncthreads();
    tile_store(sh, out, base, 0, n, 0);
}

// a[i] <- 1 / a[i], zeros stay zero.  Montgomery's trick at two levels: a lane folds its 8 elements into one product,
// groups of kInvGroup lanes fold those through LDS, and ONE Fermat inversion (255 squarings) serves 8 * kInvGroup = 64
// elements -- the inversion was 90 % of the kernel when every lane did its own.
constexpr int kInvGroup = 8;
template <int F> __global__ void __launch_bounds__(kPT) poly_batch_invert(u32 *__restrict__ a, size_t n, int canonical) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *sh = lds, *sc = lds + kPT * kPitch;          // sc: one lane product per lane
    const size_t base = (size_t)blockIdx.x * kTile;
    tile_load(sh, a, base, n, 0);
    __syncthreads();
    u32 *row = sh + threadIdx.x * kPitch;
    fe pre[kPC];
    fe run = fe_one<F>();
#pragma unroll
    for (int e = 0; e < kPC; ++e) {
        fe v = lds_get(row + 8 * e);
        if (canonical) { v = fe_to_mont<F>(v); lds_put(row + 8 * e, v); }
        pre[e] = run;
        if (!fe_is_zero(v)) run = fe_mulx<F>(run, v);
    }
    // group level: lane g of a group needs 1 / run_g = (product of the other lanes' runs) / (product of all).  The 32 group
    // inversions of a workgroup are done by the first 32 lanes, so ONE wave walks the 255-squaring loop instead of four
    // (a lane that sits idle inside a divergent loop saves nothing).
    lds_put(sc + threadIdx.x * kSPitch, run);
    __syncthreads();
    const u32 g0 = threadIdx.x & ~(u32)(kInvGroup - 1), gl = threadIdx.x & (kInvGroup - 1);
    fe before = fe_one<F>(), after = fe_one<F>();             // products of the group's runs before / after this lane
#pragma unroll
    for (int q = 0; q < kInvGroup; ++q) {
        const fe rq = lds_get(sc + (g0 + q) * kSPitch);
        if (q < (int)gl) before = fe_mulx<F>(before, rq);
        if (q > (int)gl) after = fe_mulx<F>(after, rq);
    }
    const fe others = fe_mulx<F>(before, after);
    fe total = fe_one<F>();
    if (threadIdx.x < kPT / kInvGroup) {
#pragma unroll
        for (int q = 0; q < kInvGroup; ++q) total = fe_mulx<F>(total, lds_get(sc + (threadIdx.x * kInvGroup + q) * kSPitch));
    }
    __syncthreads();
    if (threadIdx.x < kPT / kInvGroup) lds_put(sc + threadIdx.x * kSPitch, fe_inv<F>(total));
    __syncthreads();
    fe inv = fe_mulx<F>(lds_get(sc + (threadIdx.x / kInvGroup) * kSPitch), others);       // = 1 / run
#pragma unroll
    for (int e = kPC - 1; e >= 0; --e) {
        fe v = lds_get(row + 8 * e);
        if (fe_is_zero(v)) continue;
        fe r = fe_mulx<F>(inv, pre[e]);
        inv = fe_mulx<F>(inv, v);
        lds_put(row + 8 * e, canonical ? fe_from_mont<F>(r) : r);
    }
    __syncthreads();
    tile_store(sh, a, base, 0, n, 0);
}

// ---- kate division: suffix Horner scan H_i = a_i + b H_{i+1};  quotient q_{i-1} = H_i for i = 1 .. n-1 -------------
// pw[k] = b^(kPC * 2^k) for k = 0..7.  FINAL = false: agg[blk] = H at the tile's first element assuming nothing above.
template <int F, bool FINAL>
__global__ void __launch_bounds__(kPT) poly_kate_tile(const u32 *__restrict__ a, size_t n, fe b, const u32 *__restrict__ pw,
                                                      u32 *__restrict__ agg, const u32 *__restrict__ carry, u32 *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *sh = lds, *sc = lds + kPT * kPitch;
    const size_t base = (size_t)blockIdx.x * kTile;
    tile_load(sh, a, base, n, 0);
    __syncthreads();
    u32 *row = sh + threadIdx.x * kPitch;
    fe h = fe_zero();
#pragma unroll 1
    for (int e = kPC - 1; e >= 0; --e) h = fe_add<F>(lds_get(row + 8 * e), fe_mulx<F>(b, h));
    fe cin = fe_zero();
    if (FINAL) {
        cin = fe_load(carry + 8 * (size_t)blockIdx.x);
        if (threadIdx.x == kPT - 1) h = fe_add<F>(h, fe_mulx<F>(fe_load(pw), cin));   // b^kPC * H(tile end)
    }
    lds_put(sc + threadIdx.x * kSPitch, h);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int off = 1 << k;
        fe v = lds_get(sc + threadIdx.x * kSPitch);
        if ((int)threadIdx.x + off < kPT)
            v = fe_add<F>(v, fe_mulx<F>(fe_load(pw + 8 * k), lds_get(sc + (threadIdx.x + off) * kSPitch)));
        __syncthreads();
        lds_put(sc + threadIdx.x * kSPitch, v);
        __syncthreads();
    }
    if (!FINAL) {
        if (threadIdx.x == 0) fe_store(agg + 8 * (size_t)blockIdx.x, lds_get(sc));
        return;
    }
    h = threadIdx.x == kPT - 1 ? cin : lds_get(sc + (threadIdx.x + 1) * kSPitch);
#pragma unroll 1
    for (int e = kPC - 1; e >= 0; --e) {
        h = fe_add<F>(lds_get(row + 8 * e), fe_mulx<F>(b, h));
        lds_put(row + 8 * e, h);
    }
    __syncthreads();
    tile_store(sh, out, base, 1, n, 1);
}

// carry[blk] = sum_{blk' > blk} agg[blk'] m^(blk' - blk - 1), m = b^kTile; one workgroup
template <int F> __global__ void __launch_bounds__(kPT) poly_kate_carry(const u32 *__restrict__ agg, u32 nblk, fe m, u32 *__restrict__ carry) {
    __shared__ u32 sc[kPT * kSPitch];
    const u32 per = (nblk + kPT - 1) / kPT, lo = min(threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    fe h = fe_zero();
    for (u32 i = hi; i-- > lo;) h = fe_add<F>(fe_load(agg + 8 * (size_t)i), fe_mulx<F>(m, h));
    lds_put(sc + threadIdx.x * kSPitch, h);
    __syncthreads();
    fe step = fe_pow_u32<F>(m, per);
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int off = 1 << k;
        fe v = lds_get(sc + threadIdx.x * kSPitch);
        if ((int)threadIdx.x + off < kPT) v = fe_add<F>(v, fe_mulx<F>(step, lds_get(sc + (threadIdx.x + off) * kSPitch)));
        __syncthreads();
        lds_put(sc + threadIdx.x * kSPitch, v);
        step = fe_sqr<F>(step);
        __syncthreads();
    }
    h = threadIdx.x == kPT - 1 ? fe_zero() : lds_get(sc + (threadIdx.x + 1) * kSPitch);
    // h = H at position (t + 1) * per; walk down to this lane's real entries
    for (u32 i = lo + per; i-- > lo;) {
        if (i < nblk) {
            fe_store(carry + 8 * (size_t)i, h);
            h = fe_add<F>(fe_load(agg + 8 * (size_t)i), fe_mulx<F>(m, h));
        } else {
            h = fe_mulx<F>(m, h);
        }
    }
}

// ---- grand product: z[0] = init, z[i] = z[i-1] * m[i-1] ------------------------------------------------------------
template <int F, bool FINAL>
__global__ void __launch_bounds__(kPT) poly_product_tile(const u32 *__restrict__ m, size_t n, int canonical, u32 *__restrict__ agg,
                                                         const u32 *__restrict__ carry, u32 *__restrict__ z) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *sh = lds, *sc = lds + kPT * kPitch;
    const size_t base = (size_t)blockIdx.x * kTile;
    tile_load(sh, m, base, n - 1, 0);      // factors at index >= n - 1 are not part of any z
    __syncthreads();
    u32 *row = sh + threadIdx.x * kPitch;
    const fe one = fe_one<F>();
    fe p = one;
#pragma unroll 1
    for (int e = 0; e < kPC; ++e) {
        fe v = lds_get(row + 8 * e);
        const bool live = base + (size_t)threadIdx.x * kPC + e < n - 1;
        if (!live) v = one;
        else if (canonical) v = fe_to_mont<F>(v);
        lds_put(row + 8 * e, v);
        p = fe_mulx<F>(p, v);
    }
    lds_put(sc + threadIdx.x * kSPitch, p);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int off = 1 << k;
        fe v = lds_get(sc + threadIdx.x * kSPitch);
        if ((int)threadIdx.x >= off) v = fe_mulx<F>(v, lds_get(sc + (threadIdx.x - off) * kSPitch));
        __syncthreads();
        lds_put(sc + threadIdx.x * kSPitch, v);
        __syncthreads();
    }
    if (!FINAL) {
        if (threadIdx.x == kPT - 1) fe_store(agg + 8 * (size_t)blockIdx.x, lds_get(sc + (kPT - 1) * kSPitch));
        return;
    }
    fe run = fe_load(carry + 8 * (size_t)blockIdx.x);
    if (threadIdx.x) run = fe_mulx<F>(run, lds_get(sc + (threadIdx.x - 1) * kSPitch));
#pragma unroll 1
    for (int e = 0; e < kPC; ++e) {
        fe v = lds_get(row + 8 * e);
        lds_put(row + 8 * e, canonical ? fe_from_mont<F>(run) : run);
        run = fe_mulx<F>(run, v);
    }
    __syncthreads();
    tile_store(sh, z, base, 0, n, 0);
}

// carry[blk] = init * prod_{blk' < blk} agg[blk']; one workgroup
template <int F> __global__ void __launch_bounds__(kPT) poly_product_carry(const u32 *__restrict__ agg, u32 nblk, fe init, u32 *__restrict__ carry) {
    __shared__ u32 sc[kPT * kSPitch];
    const u32 per = (nblk + kPT - 1) / kPT, lo = min(threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    fe p = fe_one<F>();
    for (u32 i = lo; i < hi; ++i) p = fe_mulx<F>(p, fe_load(agg + 8 * (size_t)i));
    lds_put(sc + threadIdx.x * kSPitch, p);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int off = 1 << k;
        fe v = lds_get(sc + threadIdx.x * kSPitch);
        if ((int)threadIdx.x >= off) v = fe_mulx<F>(v, lds_get(sc + (threadIdx.x - off) * kSPitch));
        __syncthreads();
        lds_put(sc + threadIdx.x * kSPitch, v);
        __syncthreads();
    }
    fe run = init;
    if (threadIdx.x) run = fe_mulx<F>(run, lds_get(sc + (threadIdx.x - 1) * kSPitch));
    for (u32 i = lo; i < hi; ++i) {
        fe_store(carry + 8 * (size_t)i, run);
        run = fe_mulx<F>(run, fe_load(agg + 8 * (size_t)i));
    }
}

// ---- host orchestration --------------------------------------------------------------------------------------------
namespace {

struct PolyContext {
    std::mutex mu;
    DevBuf scratch, consts;
    void release_all() {
        scratch.release();
        consts.release();
    }
};
StreamContexts<PolyContext> g_poly_ctxs;
PolyContext &poly_ctx(hipStream_t st) { return g_poly_ctxs.get(st); }

}  // namespace
void poly_release_workspaces() { g_poly_ctxs.release_current_device(); }   // h2_trim
namespace {

inline fe to_fe(const u64 m[4]) {
    fe f;
    memcpy(f.v, m, 32);
    return f;
}
void host_pow(int field, u64 r[4], const u64 base[4], u64 e) {
    u64 acc[4], b[4];
    memcpy(acc, kHostField[field].one, 32);
    memcpy(b, base, 32);
    while (e) {
        if (e & 1) host_mul(field, acc, acc, b);
        host_mul(field, b, b, b);
        e >>= 1;
    }
    memcpy(r, acc, 32);
}
bool bad_field_form(int field, int form) {
    return (field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY);
}
constexpr size_t kMaxLen = (size_t)1 << 30;
constexpr unsigned kTileLds = kPT * kPitch * 4;                  // the element tile
constexpr unsigned kScanLds = kTileLds + kPT * kSPitch * 4;      // + one lane-aggregate array

// the tile kernels use more than the 64 KiB a kernel may claim without asking
int poly_kernel_attrs() {
    static std::mutex mu;
    static bool done[16];
    int dev = 0;
    H2_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev & 15]) return H2_OK;
#define H2_LDS_ATTR(k, bytes) H2_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
    H2_LDS_ATTR((poly_kate_tile<FP, false>), kScanLds);
    H2_LDS_ATTR((poly_kate_tile<FP, true>), kScanLds);
    H2_LDS_ATTR((poly_kate_tile<FQ, false>), kScanLds);
    H2_LDS_ATTR((poly_kate_tile<FQ, true>), kScanLds);
    H2_LDS_ATTR((poly_product_tile<FP, false>), kScanLds);
    H2_LDS_ATTR((poly_product_tile<FP, true>), kScanLds);
    H2_LDS_ATTR((poly_product_tile<FQ, false>), kScanLds);
    H2_LDS_ATTR((poly_product_tile<FQ, true>), kScanLds);
    H2_LDS_ATTR((poly_batch_invert<FP>), kScanLds);
    H2_LDS_ATTR((poly_batch_invert<FQ>), kScanLds);
    H2_LDS_ATTR((poly_powers<FP>), kTileLds);
    H2_LDS_ATTR((poly_powers<FQ>), kTileLds);
#undef H2_LDS_ATTR
    done[dev & 15] = true;
    return H2_OK;
}
unsigned reduce_blocks(size_t n) { return (unsigned)std::min<size_t>(1024, (n + kPT - 1) / kPT); }

#define H2_FIELD_LAUNCH(field, kern, ...)                                      \
    do {                                                                       \
        if ((field) == H2_FP) hipLaunchKernelGGL((kern<FP>), __VA_ARGS__);     \
        else hipLaunchKernelGGL((kern<FQ>), __VA_ARGS__);                      \
    } while (0)

int eval_launch(int field, const void *d_a, size_t n, const u64 *point, int form, void *d_out, hipStream_t st) {
    PolyContext &cx = poly_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if (n == 0) { H2_HIP(hipMemsetAsync(d_out, 0, 32, st)); return H2_OK; }
    // every lane pays x^t (square and multiply, ~27 multiplications at 2^18 lanes) on top of its Horner steps: 65536 lanes x 16
    // coefficients keep that a third of the work (1024 blocks x 4 coefficients: 87 % of it; 79 -> ~35 us at n = 2^20)
    const unsigned blocks = std::min(256u, reduce_blocks(n));
    int rc = cx.scratch.reserve((size_t)blocks * 32);
    if (rc != H2_OK) return rc;
    u64 x[4], xT[4];
    host_to_mont(field, x, point, form);
    host_pow(field, xT, x, (u64)blocks * kPT);
    // linear in the coefficients: canonical coefficients with a Montgomery point give the canonical value
    H2_FIELD_LAUNCH(field, poly_eval_partial, dim3(blocks), dim3(kPT), 0, st, (const u32 *)d_a, n, to_fe(x), to_fe(xT), cx.scratch.as<u32>());
    H2_FIELD_LAUNCH(field, poly_sum_partials, dim3(1), dim3(kPT), 0, st, cx.scratch.as<u32>(), blocks, 0, (u32 *)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int inner_launch(int field, const void *d_a, const void *d_b, size_t n, int form, void *d_out, hipStream_t st) {
    PolyContext &cx = poly_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if (n == 0) { H2_HIP(hipMemsetAsync(d_out, 0, 32, st)); return H2_OK; }
    const unsigned blocks = reduce_blocks(n);
    int rc = cx.scratch.reserve((size_t)blocks * 32);
    if (rc != H2_OK) return rc;
    H2_FIELD_LAUNCH(field, poly_inner_partial, dim3(blocks), dim3(kPT), 0, st, (const u32 *)d_a, (const u32 *)d_b, n, cx.scratch.as<u32>());
    H2_FIELD_LAUNCH(field, poly_sum_partials, dim3(1), dim3(kPT), 0, st, cx.scratch.as<u32>(), blocks, form == H2_FORM_CANONICAL ? 1 : 0,
                    (u32 *)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int kate_launch(int field, const void *d_a, size_t n, const u64 *point, int form, void *d_out, hipStream_t st) {
    if (n <= 1) return H2_OK;
    int rc = poly_kernel_attrs();
    if (rc != H2_OK) return rc;
    PolyContext &cx = poly_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const unsigned nblk = (unsigned)((n + kTile - 1) / kTile);
    if ((rc = cx.scratch.reserve((size_t)nblk * 64)) != H2_OK) return rc;
    if ((rc = cx.consts.reserve(8 * 32)) != H2_OK) return rc;
    u64 b[4], pw[8][4], m[4];
    host_to_mont(field, b, point, form);
    host_pow(field, pw[0], b, kPC);
    for (int k = 1; k < 8; ++k) host_mul(field, pw[k], pw[k - 1], pw[k - 1]);
    host_mul(field, m, pw[7], pw[7]);      // b^(kPC * 256) = b^kTile
    // the constant table belongs to this (device, stream): the copy is stream-ordered after the kernels that last read it
    H2_HIP(hipMemcpyAsync(cx.consts.ptr, pw, sizeof(pw), hipMemcpyHostToDevice, st));
    u32 *agg = cx.scratch.as<u32>(), *carry = agg + 8 * (size_t)nblk;
    // linear in the coefficients, so canonical and Montgomery inputs take the same path
    if (field == H2_FP) {
        hipLaunchKernelGGL((poly_kate_tile<FP, false>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_a, n, to_fe(b), cx.consts.as<u32>(), agg, carry, (u32 *)d_out);
        hipLaunchKernelGGL((poly_kate_carry<FP>), dim3(1), dim3(kPT), 0, st, agg, nblk, to_fe(m), carry);
        hipLaunchKernelGGL((poly_kate_tile<FP, true>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_a, n, to_fe(b), cx.consts.as<u32>(), agg, carry, (u32 *)d_out);
    } else {
        hipLaunchKernelGGL((poly_kate_tile<FQ, false>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_a, n, to_fe(b), cx.consts.as<u32>(), agg, carry, (u32 *)d_out);
        hipLaunchKernelGGL((poly_kate_carry<FQ>), dim3(1), dim3(kPT), 0, st, agg, nblk, to_fe(m), carry);
        hipLaunchKernelGGL((poly_kate_tile<FQ, true>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_a, n, to_fe(b), cx.consts.as<u32>(), agg, carry, (u32 *)d_out);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int product_launch(int field, const void *d_m, size_t n, const u64 *init, int form, void *d_z, hipStream_t st) {
    if (n == 0) return H2_OK;
    int rc = poly_kernel_attrs();
    if (rc != H2_OK) return rc;
    PolyContext &cx = poly_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const unsigned nblk = (unsigned)((n + kTile - 1) / kTile);
    if ((rc = cx.scratch.reserve((size_t)nblk * 64)) != H2_OK) return rc;
    u64 i0[4];
    host_to_mont(field, i0, init, form);
    const int canonical = form == H2_FORM_CANONICAL;
    u32 *agg = cx.scratch.as<u32>(), *carry = agg + 8 * (size_t)nblk;
    if (field == H2_FP) {
        hipLaunchKernelGGL((poly_product_tile<FP, false>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_m, n, canonical, agg, carry, (u32 *)d_z);
        hipLaunchKernelGGL((poly_product_carry<FP>), dim3(1), dim3(kPT), 0, st, agg, nblk, to_fe(i0), carry);
        hipLaunchKernelGGL((poly_product_tile<FP, true>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_m, n, canonical, agg, carry, (u32 *)d_z);
    } else {
        hipLaunchKernelGGL((poly_product_tile<FQ, false>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_m, n, canonical, agg, carry, (u32 *)d_z);
        hipLaunchKernelGGL((poly_product_carry<FQ>), dim3(1), dim3(kPT), 0, st, agg, nblk, to_fe(i0), carry);
        hipLaunchKernelGGL((poly_product_tile<FQ, true>), dim3(nblk), dim3(kPT), kScanLds, st, (const u32 *)d_m, n, canonical, agg, carry, (u32 *)d_z);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// host-pointer staging: device copies of the inputs, released on scope exit
struct Staged {
    std::vector<void *> bufs;
    ~Staged() {
        for (void *p : bufs) (void)hipFree(p);
    }
    int alloc(size_t bytes, void **d) {
        H2_HIP(hipMalloc(d, bytes ? bytes : 32));
        bufs.push_back(*d);
        return H2_OK;
    }
    int up(const void *h, size_t bytes, void **d) {
        int rc = alloc(bytes, d);
        if (rc != H2_OK) return rc;
        if (bytes) H2_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return H2_OK;
    }
};

}  // namespace
}  // namespace h2

using namespace h2;

extern "C" int h2_eval_polynomial_device(int field, const void *d_poly, size_t n, const uint64_t *point, int form, void *d_out, void *stream) {
    if (bad_field_form(field, form) || !point || !d_out || (n && !d_poly) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return eval_launch(field, d_poly, n, point, form, d_out, (hipStream_t)stream);
}

extern "C" int h2_eval_polynomial(int field, const uint64_t *poly, size_t n, const uint64_t *point, int form, uint64_t *out) {
    if (bad_field_form(field, form) || !point || !out || (n && !poly) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    Staged s;
    void *d_a, *d_o;
    if ((rc = s.up(poly, n * 32, &d_a)) != H2_OK || (rc = s.alloc(32, &d_o)) != H2_OK) return rc;
    if ((rc = eval_launch(field, d_a, n, point, form, d_o, 0)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(out, d_o, 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_inner_product_device(int field, const void *d_a, const void *d_b, size_t n, int form, void *d_out, void *stream) {
    if (bad_field_form(field, form) || !d_out || (n && (!d_a || !d_b)) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return inner_launch(field, d_a, d_b, n, form, d_out, (hipStream_t)stream);
}

extern "C" int h2_inner_product(int field, const uint64_t *a, const uint64_t *b, size_t n, int form, uint64_t *out) {
    if (bad_field_form(field, form) || !out || (n && (!a || !b)) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    Staged s;
    void *d_a, *d_b, *d_o;
    if ((rc = s.up(a, n * 32, &d_a)) != H2_OK || (rc = s.up(b, n * 32, &d_b)) != H2_OK || (rc = s.alloc(32, &d_o)) != H2_OK) return rc;
    if ((rc = inner_launch(field, d_a, d_b, n, form, d_o, 0)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(out, d_o, 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_kate_division_device(int field, const void *d_a, size_t n, const uint64_t *point, int form, void *d_out, void *stream) {
    if (bad_field_form(field, form) || !point || n == 0 || !d_a || (n > 1 && !d_out) || n > kMaxLen || d_a == d_out) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return kate_launch(field, d_a, n, point, form, d_out, (hipStream_t)stream);
}

extern "C" int h2_kate_division(int field, const uint64_t *a, size_t n, const uint64_t *point, int form, uint64_t *out) {
    if (bad_field_form(field, form) || !point || n == 0 || !a || (n > 1 && !out) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (n == 1) return H2_OK;
    Staged s;
    void *d_a, *d_o;
    if ((rc = s.up(a, n * 32, &d_a)) != H2_OK || (rc = s.alloc((n - 1) * 32, &d_o)) != H2_OK) return rc;
    if ((rc = kate_launch(field, d_a, n, point, form, d_o, 0)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(out, d_o, (n - 1) * 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_powers_device(int field, const uint64_t *x, size_t n, int form, void *d_out, void *stream) {
    if (bad_field_form(field, form) || !x || (n && !d_out) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    if ((rc = poly_kernel_attrs()) != H2_OK) return rc;
    u64 xm[4];
    host_to_mont(field, xm, x, form);
    H2_FIELD_LAUNCH(field, poly_powers, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kPT), kTileLds, (hipStream_t)stream, (u32 *)d_out, n, to_fe(xm),
                    form == H2_FORM_CANONICAL ? 1 : 0);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_powers(int field, const uint64_t *x, size_t n, int form, uint64_t *out) {
    if (bad_field_form(field, form) || !x || (n && !out) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    Staged s;
    void *d_o;
    if ((rc = s.alloc(n * 32, &d_o)) != H2_OK) return rc;
    if ((rc = h2_powers_device(field, x, n, form, d_o, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(out, d_o, n * 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_scale_add_device(int field, void *d_a, const uint64_t *x, const void *d_b, size_t n, int form, void *stream) {
    if (bad_field_form(field, form) || !x || (n && (!d_a || !d_b)) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    u64 xm[4];
    host_to_mont(field, xm, x, form);      // a Montgomery factor keeps a and b in whichever form they are
    H2_FIELD_LAUNCH(field, poly_scale_add, dim3((unsigned)((n + kPT - 1) / kPT)), dim3(kPT), 0, (hipStream_t)stream, (u32 *)d_a, (const u32 *)d_b, n,
                    to_fe(xm));
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_scale_add(int field, uint64_t *a, const uint64_t *x, const uint64_t *b, size_t n, int form) {
    if (bad_field_form(field, form) || !x || (n && (!a || !b)) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    Staged s;
    void *d_a, *d_b;
    if ((rc = s.up(a, n * 32, &d_a)) != H2_OK || (rc = s.up(b, n * 32, &d_b)) != H2_OK) return rc;
    if ((rc = h2_scale_add_device(field, d_a, x, d_b, n, form, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(a, d_a, n * 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_batch_invert_device(int field, void *d_a, size_t n, int form, void *stream) {
    if (bad_field_form(field, form) || (n && !d_a) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    if ((rc = poly_kernel_attrs()) != H2_OK) return rc;
    H2_FIELD_LAUNCH(field, poly_batch_invert, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kPT), kScanLds, (hipStream_t)stream, (u32 *)d_a, n,
                    form == H2_FORM_CANONICAL ? 1 : 0);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_batch_invert(int field, uint64_t *a, size_t n, int form) {
    if (bad_field_form(field, form) || (n && !a) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    Staged s;
    void *d_a;
    if ((rc = s.up(a, n * 32, &d_a)) != H2_OK) return rc;
    if ((rc = h2_batch_invert_device(field, d_a, n, form, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(a, d_a, n * 32, hipMemcpyDeviceToHost));
    return H2_OK;
}

extern "C" int h2_grand_product_device(int field, const void *d_m, size_t n, const uint64_t *init, int form, void *d_z, void *stream) {
    if (bad_field_form(field, form) || !init || (n && !d_z) || (n > 1 && !d_m) || n > kMaxLen || (n && d_m == d_z)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return product_launch(field, d_m, n, init, form, d_z, (hipStream_t)stream);
}

extern "C" int h2_grand_product(int field, const uint64_t *m, size_t n, const uint64_t *init, int form, uint64_t *z) {
    if (bad_field_form(field, form) || !init || (n && !z) || (n > 1 && !m) || n > kMaxLen) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    Staged s;
    void *d_m, *d_z;
    if ((rc = s.up(m, n > 1 ? (n - 1) * 32 : 0, &d_m)) != H2_OK || (rc = s.alloc(n * 32, &d_z)) != H2_OK) return rc;
    if ((rc = product_launch(field, d_m, n, init, form, d_z, 0)) != H2_OK) return rc;
    H2_HIP(hipMemcpy(z, d_z, n * 32, hipMemcpyDeviceToHost));
    return H2_OK;
}
