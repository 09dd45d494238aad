// Registered bases (Params::g / g_lagrange, poly/commitment.rs:26-33): the precomputed table [W][n + 1] of 2^(c w) P_i in M9 form, the blind
// base column, and the handle bookkeeping behind h2_bases_*.
#include "msm_internal.cuh"

namespace h2 {

// ---- precomputed table for registered bases: row w holds 2^(c*w) * P_i as affine points --------------
// chain: one lane per point walks w = 1 .. W-1 with c doublings each, parking XYZZ in `tmp`
template <int FB>
__global__ void __launch_bounds__(256) msm_table_chain(const u32 *__restrict__ row0, u32 *__restrict__ tmp, u32 count,
                                                       u32 first, u32 stride, int c, int W) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    affine<FB> p = aff_load<FB>(row0 + 16 * (size_t)(first + i));
    xyzz<FB> r = xyzz_identity<FB>();
    xyzz_madd<FB>(r, p);
    for (int w = 1; w < W; ++w) {
        for (int k = 0; k < c; ++k) r = xyzz_dbl<FB>(r);
        xyzz_store<FB>(tmp + 32 * ((size_t)(w - 1) * count + i), r);
    }
    (void)stride;
}
// the same chain with one point per quad of lanes (curve_wide.cuh): small tables are bound by the (W - 1) c sequential doublings
template <int FB>
__global__ void __launch_bounds__(256) msm_table_chain_wide(const u32 *__restrict__ row0, u32 *__restrict__ tmp, u32 count,
                                                            u32 first, int c, int W) {
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (i >= count) return;
    const affine<FB> p = aff_load<FB>(row0 + 16 * (size_t)(first + i));
    xyzz<FB> r = xyzz_identity<FB>();
    xyzz_madd<FB>(r, p);
    xyzz9<FB> r9 = xyzz9_from_r256_wide<FB>(r);              // the chain itself on the carry-free layer (curve9_wide.cuh)
    for (int w = 1; w < W; ++w) {
        for (int k = 0; k < c; ++k) r9 = xyzz9_dbl_wide<FB>(r9);
        const xyzz<FB> out = xyzz9_to_r256_wide<FB>(r9);
        if ((threadIdx.x & (kGroup - 1)) == 0) xyzz_store<FB>(tmp + 32 * ((size_t)(w - 1) * count + i), out);
    }
}
// blind base: column `col` of the table must hold the multiples of `w` (Params::w, poly/commitment.rs:26-33).  ONE workgroup of
// 64 lanes: lane 0 compares w with what row 0 of the column holds (the CONTENT, not an address) and leaves when they agree.
// Otherwise it walks the doubling chain, parking 2^(c w) * w in LDS, lanes 1 .. W-1 normalise one row each, and row 0 -- the
// word later calls compare against -- is written LAST, after a fence: a column whose row 0 shows w is complete.
template <int FB>
__global__ void __launch_bounds__(64) msm_blind_install(u32 *__restrict__ table, const u32 *__restrict__ w_xy, u32 col, u32 stride, int c, int W,
                                                        int mont) {
    __shared__ __attribute__((aligned(16))) u32 chain[63 * 32];
    __shared__ u32 differs;
    affine<FB> p9;
    if (threadIdx.x == 0) {
        affine<FB> p = aff_load<FB>(w_xy);
        if (!mont) { p.x = fe_to_mont<FB>(p.x); p.y = fe_to_mont<FB>(p.y); }
        p9 = aff_to_m9<FB>(p);                                                                   // the table holds M9 form
        const affine<FB> cur = aff_load<FB>(table + 16 * (size_t)col);
        differs = (fe_eq(p9.x, cur.x) && fe_eq(p9.y, cur.y)) ? 0u : 1u;
        if (differs) {
            xyzz<FB> r = xyzz_identity<FB>();
            xyzz_madd<FB>(r, p);
            for (int w = 1; w < W; ++w) {
                for (int k = 0; k < c; ++k) r = xyzz_dbl<FB>(r);
                xyzz_store<FB>(chain + 32 * (size_t)(w - 1), r);
            }
        }
    }
    __syncthreads();
    if (!differs) return;
    const u32 w = threadIdx.x;
    if (w >= 1 && (int)w < W) {
        const xyzz<FB> r = xyzz_load<FB>(chain + 32 * (size_t)(w - 1));
        const affine<FB> a = aff_to_m9<FB>(xyzz_to_affine<FB>(r));
        u32 *dst = table + 16 * ((size_t)w * stride + col);
        fe_store(dst, a.x);
        fe_store(dst + 8, a.y);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        fe_store(table + 16 * (size_t)col, p9.x);
        fe_store(table + 16 * (size_t)col + 8, p9.y);
    }
}

// rows 1 .. W-1 of the table from the chains' XYZZ results, affine and in M9 form, with ONE inversion per point: the W - 1
// multiples of a point are normalised together (Montgomery's trick over
// d_w = ZZ_w ZZZ_w; the running products wait in `pre`), ~8 multiplications per entry instead of a 255-step inversion each
// phi_rows != 0 (endomorphism tables): row phi_rows + w receives phi of what row w receives -- one more product per entry
template <int FB>
__global__ void __launch_bounds__(256) msm_table_normalise_batch(const u32 *__restrict__ tmp, u32 *__restrict__ pre, u32 *__restrict__ table,
                                                                 u32 count, u32 first, u32 stride, int W, int phi_rows = 0) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe acc = fe_one<FB>();
    for (int w = 1; w < W; ++w) {
        const size_t t = (size_t)(w - 1) * count + i;
        const fe zz = fe_load(tmp + 32 * t + 16), zzz = fe_load(tmp + 32 * t + 24);
        fe_store(pre + 8 * t, acc);
        if (!fe_is_zero(zz)) acc = fe_mulx<FB>(acc, fe_mulx<FB>(zz, zzz));       // the identity (exact zeros) sits the product out
    }
    fe inv = fe_inv<FB>(acc);
    for (int w = W - 1; w >= 1; --w) {
        const size_t t = (size_t)(w - 1) * count + i;
        const xyzz<FB> r = xyzz_load<FB>(tmp + 32 * t);
        u32 *dst = table + 16 * ((size_t)w * stride + first + i);
        if (fe_is_zero(r.zz)) {
            fe_store(dst, fe_zero());
            fe_store(dst + 8, fe_zero());
            continue;
        }
        const fe di = fe_mulx<FB>(inv, fe_load(pre + 8 * t));                      // 1 / (ZZ ZZZ)
        inv = fe_mulx<FB>(inv, fe_mulx<FB>(r.zz, r.zzz));
        const affine<FB> am = affine<FB>{fe_mulx<FB>(r.x, fe_mulx<FB>(di, r.zzz)), fe_mulx<FB>(r.y, fe_mulx<FB>(di, r.zz))};
        const affine<FB> a = aff_to_m9<FB>(am);
        fe_store(dst, a.x);
        fe_store(dst + 8, a.y);
        if (phi_rows) {
            const affine<FB> ph = aff_to_m9<FB>(affine<FB>{fe_mulx<FB>(am.x, glv_zeta<FB>()), am.y});
            u32 *dph = dst + 16 * (size_t)phi_rows * stride;
            fe_store(dph, ph.x);
            fe_store(dph + 8, ph.y);
        }
    }
}
// row 0 (the caller's points, reference Montgomery form) -> M9 form, once the chains have read it
template <int FB>
__global__ void __launch_bounds__(256) msm_table_row0_to_m9(u32 *__restrict__ table, u32 count, u32 first, u32 phi_row_words = 0) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u32 *dst = table + 16 * (size_t)(first + i);
    const affine<FB> am = aff_load<FB>(dst);
    const affine<FB> a = aff_to_m9<FB>(am);
    fe_store(dst, a.x);
    fe_store(dst + 8, a.y);
    if (phi_row_words && !aff_is_identity(am)) {       // (endomorphism tables: row `glv` = phi(row 0); the identity stays all-zero)
        const affine<FB> ph = aff_to_m9<FB>(affine<FB>{fe_mulx<FB>(am.x, glv_zeta<FB>()), am.y});
        fe_store(dst + phi_row_words, ph.x);
        fe_store(dst + phi_row_words + 8, ph.y);
    }
}

static std::mutex g_bases_mu;
static std::map<h2_bases_t, std::shared_ptr<Bases>> g_bases;
static h2_bases_t g_next_handle = 1;

// endomorphism: may the handle be an ENDOMORPHISM table (Bases::glv)?  Such a table belongs to the opening argument's round loop and never leaves the
// library; only the sub-digit paired commit, its refill and the bookkeeping entry points read it -- to everything else it is not a handle.
std::shared_ptr<Bases> find_bases(h2_bases_t h, bool endomorphism) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(h);
    if (it == g_bases.end() || (it->second->glv && !endomorphism)) return nullptr;
    return it->second;
}

bool bad_common(int curve, int form, int out_kind) {
    return (curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) ||
           (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE);
}

// fills rows 1..W-1 of the table for columns [first, first + count) from row 0
// keep_tmp: the staging stays with the handle (a refilled table pays hipMalloc / hipFree -- ~0.3 ms, and a device synchronisation each -- once)
int table_fill(Bases &b, u32 first, u32 count, hipStream_t st, bool keep_tmp) {
    if (!count || b.W <= 1) return H2_OK;
    void *tmp = nullptr;
    // worked in slabs so the XYZZ staging stays modest
    const u32 slab = 1u << 18;
    const int Wd = b.glv ? b.glv : b.W;              // rows that come from the doubling chain (an endomorphism table: half of them, 8 x 16 doublings)
    const size_t tmp_bytes = (size_t)std::min(count, slab) * (Wd - 1) * 160;      // XYZZ staging + the running products
    keep_tmp = keep_tmp && tmp_bytes <= ((size_t)256 << 20);
    if (keep_tmp) {
        int rc = b.fill_tmp.reserve(tmp_bytes);
        if (rc != H2_OK) return rc;
        tmp = b.fill_tmp.ptr;
    } else {
        H2_HIP(hipMalloc(&tmp, tmp_bytes));
    }
    for (u32 off = 0; off < count; off += slab) {
        u32 cnt = std::min(slab, count - off);
        dim3 g1((cnt + 255) / 256), blk(256);
        size_t tot = (size_t)cnt * (Wd - 1);
        u32 *pre = (u32 *)tmp + 32 * tot;
        const bool wide = count <= 65536;           // few points: the doubling chain is pure latency
        dim3 g1w((cnt * kGroup + 255) / 256);
        if (b.curve == H2_PALLAS) {
            if (wide) hipLaunchKernelGGL((msm_table_chain_wide<FP>), g1w, blk, 0, st, (const u32 *)b.d_table, (u32 *)tmp, cnt, first + off, b.c, Wd);
            else hipLaunchKernelGGL((msm_table_chain<FP>), g1, blk, 0, st, (const u32 *)b.d_table, (u32 *)tmp, cnt, first + off, b.stride, b.c, Wd);
            hipLaunchKernelGGL((msm_table_normalise_batch<FP>), g1, blk, 0, st, (const u32 *)tmp, pre, (u32 *)b.d_table, cnt, first + off, b.stride, Wd, b.glv);
        } else {
            if (wide) hipLaunchKernelGGL((msm_table_chain_wide<FQ>), g1w, blk, 0, st, (const u32 *)b.d_table, (u32 *)tmp, cnt, first + off, b.c, Wd);
            else hipLaunchKernelGGL((msm_table_chain<FQ>), g1, blk, 0, st, (const u32 *)b.d_table, (u32 *)tmp, cnt, first + off, b.stride, b.c, Wd);
            hipLaunchKernelGGL((msm_table_normalise_batch<FQ>), g1, blk, 0, st, (const u32 *)tmp, pre, (u32 *)b.d_table, cnt, first + off, b.stride, Wd, b.glv);
        }
    }
    {
        dim3 g0((count + 255) / 256), blk(256);
        const u32 phi_words = b.glv ? 16u * (u32)b.glv * b.stride : 0u;
        if (b.curve == H2_PALLAS) hipLaunchKernelGGL((msm_table_row0_to_m9<FP>), g0, blk, 0, st, (u32 *)b.d_table, count, first, phi_words);
        else hipLaunchKernelGGL((msm_table_row0_to_m9<FQ>), g0, blk, 0, st, (u32 *)b.d_table, count, first, phi_words);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (!keep_tmp) (void)hipFree(tmp);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---- the blind base ---------------------------------------------------------------------------------------------------
// `Params::w` is a field of `Params`, fixed for its life (poly/commitment.rs:26-33, set at :102-103), so it is a property of
// the HANDLE: h2_bases_set_blind_base installs the multiples of w as column n of the table once, and a commit that passes a
// blind scalar but no w uses that column -- no kernel, no event, nothing that ties the commits of different streams together.
// A commit may still present a w of its own: the 64 BYTES are compared (never an address) -- on the host for host pointers,
// by msm_blind_install on the commit's stream for device pointers -- and only a different point rebuilds the column.
static void launch_blind_install(Bases &b, const void *d_w_xy, int form, hipStream_t st) {
    if (b.curve == H2_PALLAS)
        hipLaunchKernelGGL((msm_blind_install<FP>), dim3(1), dim3(64), 0, st, (u32 *)b.d_table, (const u32 *)d_w_xy, (u32)b.n, b.stride, b.c, b.W,
                           form == H2_FORM_MONTGOMERY);
    else
        hipLaunchKernelGGL((msm_blind_install<FQ>), dim3(1), dim3(64), 0, st, (u32 *)b.d_table, (const u32 *)d_w_xy, (u32)b.n, b.stride, b.c, b.W,
                           form == H2_FORM_MONTGOMERY);
}

// host pointer: compared by content against what the handle was last given; a different point waits for the device to drain
// (commits with the old w may be in flight on any stream), installs the new one and returns when the column is complete.
int set_blind_base_host(Bases &b, const void *host_w_xy, int form) {
    std::lock_guard<std::mutex> lk(b.mu);
    if (b.blind_set && b.blind_host_known && b.blind_form == form && memcmp(b.blind_host, host_w_xy, 64) == 0) return H2_OK;
    int cur = 0;
    H2_HIP(hipGetDevice(&cur));
    if (cur != b.device) H2_HIP(hipSetDevice(b.device));
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(b.d_blind_tmp, host_w_xy, 64, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_blind_install(b, b.d_blind_tmp, form, 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (cur != b.device) (void)hipSetDevice(cur);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
    b.blind_set = b.blind_host_known = true;
    b.blind_form = form;
    memcpy(b.blind_host, host_w_xy, 64);
    return H2_OK;
}

// device pointer presented by a commit: one 64-lane kernel on the commit's stream compares the bytes with row 0 of the column
// and rebuilds it only when they differ (then this w becomes the handle's).  Commits that use another w on other streams
// must have completed by then, as for any change of `Params`.
int override_blind_base_device(Bases &b, const void *d_w_xy, int form, hipStream_t st) {
    std::lock_guard<std::mutex> lk(b.mu);
    launch_blind_install(b, d_w_xy, form, st);
    H2_HIP(hipGetLastError());
    b.blind_set = true;
    b.blind_host_known = false;
    return H2_OK;
}


}  // namespace h2

using namespace h2;

static int bases_register_impl(int curve, const void *bases_xy, bool on_device, size_t n, int form, h2_bases_t *handle, int want_c = 0, bool glv = false) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) ||
        !handle || (n && !bases_xy) || n > (1u << 26))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    auto b = std::make_shared<Bases>();
    b->curve = curve;
    b->n = n;
    b->c = want_c ? want_c : choose_c(n ? n : 1, true);
    b->W = 255 / b->c + 1;
    if (glv) {                 // the halves of a split scalar have 129 bits: nine 16-bit windows each, the second set of rows through phi
        b->c = 16;
        b->glv = 9;
        b->W = 18;
    }
    b->stride = (u32)n + 1;
    H2_HIP(hipGetDevice(&b->device));
    H2_HIP(hipMalloc(&b->d_table, (size_t)b->W * b->stride * 64));
    H2_HIP(hipMemsetAsync(b->d_table, 0, (size_t)b->W * b->stride * 64, 0));
    H2_HIP(hipMalloc(&b->d_blind_tmp, 64));
    if (n) {
        H2_HIP(hipMemcpyAsync(b->d_table, bases_xy, n * 64, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, 0));
        if (form == H2_FORM_CANONICAL) to_mont_async(curve, (u32 *)b->d_table, n * 2, 0);
        if ((rc = table_fill(*b, 0, (u32)n, 0)) != H2_OK) return rc;   // ~Bases releases the allocations
    }
    H2_HIP(hipStreamSynchronize(0));
    std::lock_guard<std::mutex> lk(g_bases_mu);
    h2_bases_t h = g_next_handle++;
    g_bases[h] = b;
    *handle = h;
    return H2_OK;
}

extern "C" int h2_bases_register(int curve, const uint64_t *bases_xy, size_t n, int form, h2_bases_t *handle) {
    return bases_register_impl(curve, bases_xy, false, n, form, handle);
}

// Window width for a table that only serves independent column commits (Params::g, g_lagrange): from 2^18 points on 17 bits --
// 255 = 15 x 17, so a scalar leaves 15 digits instead of 16 (the top window of a scalar below q never exceeds 2^16, half the
// window, so the signed recode carries nothing out of it): the accumulate is ~6 % shorter, the sort and
// the fold (2^16 buckets) ~0.07 ms longer, which independent commits hide (953-966 against 925-939 M scalar-mults/s, one box,
// one lone commit unchanged).  The paired commit and the collapsed-generator read-out of the opening argument take 16-bit
// tables, which is what h2_bases_register keeps building.
extern "C" int h2_commit_column_window_bits(size_t n) {
    const int c = choose_c(n ? n : 1, true);
    int lowb, lb;
    if (const char *e = ab_env("H2_COLUMN_C")) {          // sweeps only (bench.py, bench/tools): the width column tables are built with
        const int v = atoi(e);
        if (v >= 4 && v <= kMaxCShared && (v <= kMaxC || (n + 1 < ((size_t)1 << 31) && sort2_geometry((u32)n + 1, v, &lowb, &lb)))) return v;
    }
    return c == 16 && n >= ((size_t)1 << 18) && n + 1 < ((size_t)1 << 31) && sort2_geometry((u32)n + 1, 17, &lowb, &lb) ? 17 : c;
}

extern "C" int h2_bases_register_ex(int curve, const uint64_t *bases_xy, size_t n, int form, int window_bits, h2_bases_t *handle) {
    if (window_bits) {
        int lowb, lb;
        if (window_bits < 4 || window_bits > kMaxCShared ||
            (window_bits > kMaxC && !(n + 1 < ((size_t)1 << 31) && sort2_geometry((u32)n + 1, window_bits, &lowb, &lb))))
            return H2_ERR_ARGS;
    }
    return bases_register_impl(curve, bases_xy, false, n, form, handle, window_bits);
}

// the same from points already in HBM (work queued on other streams that produces them must have completed: the copy runs on
// the null stream).  What the opening argument registers its collapsed generators with (h2_ipa_collapsed_generators_device).
extern "C" int h2_bases_register_device(int curve, const void *d_bases_xy, size_t n, int form, h2_bases_t *handle) {
    return bases_register_impl(curve, d_bases_xy, true, n, form, handle);
}

// Internal (ipa.hip): rebuild the table of an existing handle from n new points in HBM -- same n, same window width, the allocation
// is kept.  The opening argument registers a table for its collapsed generators in every proof; from the second proof on this
// saves the allocation and the release (~0.35 ms of hipMalloc / hipFree, which also synchronise the device).  The handle's blind
// column is cleared with the table.  Nothing else may be using the handle (the caller owns it).
namespace h2 {
// Does h2_commit_pair_device take the sub-digit form for a table of n points (16-bit windows)?  (H2_PAIR_SUBDIGITS: 0 = never; n = the largest table.)
static long pair_subdigit_max() {
    static const long v = [] { const char *e = ab_env("H2_PAIR_SUBDIGITS"); return e ? atol(e) : (long)((1 << 16) + 4); }();
    return v;
}
bool pair_subdigits_apply(size_t n) { return pair_subdigit_max() > 0 && n >= 8 && n <= (size_t)pair_subdigit_max(); }
// Internal (ipa.hip): the table of the opening argument's collapsed generators.  glv: an ENDOMORPHISM table (Bases::glv) -- nine rows by the doubling
// chain instead of sixteen (128 dependent doublings instead of 240: the chain is the latency of the switch), nine more through phi; only the
// sub-digit paired commit reads such a table.
int bases_register_device_internal(int curve, const void *d_bases_xy, size_t n, int form, h2_bases_t *handle, bool glv) {
    return bases_register_impl(curve, d_bases_xy, true, n, form, handle, 0, glv);
}
int bases_refill_device(h2_bases_t handle, const void *d_bases_xy, size_t n, int form) {
    auto b = find_bases(handle, true);
    if (!b || b->n != n || !d_bases_xy || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> bl(b->mu);
    H2_HIP(hipMemsetAsync(b->d_table, 0, (size_t)b->W * b->stride * 64, 0));
    H2_HIP(hipMemcpyAsync(b->d_table, d_bases_xy, n * 64, hipMemcpyDeviceToDevice, 0));
    if (form == H2_FORM_CANONICAL) to_mont_async(b->curve, (u32 *)b->d_table, n * 2, 0);
    b->blind_set = b->blind_host_known = false;
    if ((rc = table_fill(*b, 0, (u32)n, 0, true)) != H2_OK) return rc;
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}
}  // namespace h2

extern "C" int h2_bases_info(h2_bases_t handle, size_t *n, int *window_bits, int *curve) {
    auto b = find_bases(handle, true);
    if (!b) return H2_ERR_HANDLE;
    if (n) *n = b->n;
    if (window_bits) *window_bits = b->c;
    if (curve) *curve = b->curve;
    return H2_OK;
}

extern "C" int h2_bases_blind_base_set(h2_bases_t handle) {
    auto b = find_bases(handle);
    if (!b) return -H2_ERR_HANDLE;
    std::lock_guard<std::mutex> bl(b->mu);
    return b->blind_set ? 1 : 0;
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
    b.reset();   // frees now unless a concurrent commit still holds a reference (then when that call returns)
    return H2_OK;
}

