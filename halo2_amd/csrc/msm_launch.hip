// Pippenger multi-scalar multiplication for Pallas / Vesta on gfx950.
//
// Replaces the body of `best_multiexp` (halo2_proofs/src/arithmetic.rs:143-180) and `Buckets::sum`
// (:74-93), and -- for bases registered once per `Params` -- `Params::commit` / `commit_lagrange`
// (poly/commitment.rs:119-150).  The reference runs one CPU task per c-bit window, each re-streaming
// all n (scalar, base) pairs; the result is a group element, so window width, digit encoding and
// summation order are free.  Organised for the GPU instead:
//
//   recode      one lane per scalar: Montgomery -> canonical once (the reference redoes `to_repr` per
//               window, :77), W signed c-bit digits -> u16 codes, window-major                 [HBM]
//   count       per (slice, chunk) workgroup: 2^(c-1)-bin histogram in LDS (128 KiB at c = 16; global
//               atomics measured 20x slower), written out as per-chunk slices                   [LDS]
//   scan        chunk prefixes, then a three-kernel exclusive scan giving every bucket its entry
//               offset and the offsets of its fixed-size work parts
//   scatter     same workgroups: offsets in LDS, LDS atomics hand out slots; base index | sign
//               lands in the bucket-sorted entry list                                           [LDS]
//   accumulate  the hot kernel: the sorted entry list is cut into T EQUAL ranges, T = the lanes the chip
//               keeps resident; each lane gathers 64-B affine bases and does XYZZ mixed additions in
//               registers, flushing a segment whenever its range crosses a bucket boundary.  Every lane
//               does the same work whatever the digit distribution (repeated or tiny scalars)      [VALU]
//   finish      bucket = its own segment + the heads of the ranges that begin inside it
//   reduce      running-sum fold (:86-92) restructured as 8-bucket segments + a small scalar
//               multiple per segment, then a tree sum per slice
//   combine     Horner over windows (:169-178) on one lane; emits Jacobian or affine
//
// Two shapes share these kernels:
//   generic     (h2_msm) W slices of 2^(c-1) buckets, one per window; combine does the c*i doublings.
//   registered  (h2_bases_register / h2_commit) the table [W][n+1] of 2^(c*w) * P_i is precomputed in
//               HBM (1 GiB at k = 20 -- nothing next to 288 GB), so all windows share ONE slice of
//               buckets: 16x fewer buckets to reduce and no doubling chain at all.  Column n holds the
//               blind's base `w` (poly/commitment.rs:127).
//
// No MFMA anywhere: this is modular-integer arithmetic.  Bound by VALU integer-multiply issue, not
// HBM: algorithmic traffic is 96 B per (scalar, base) pair against ~1.8e2 modular multiplies.
#include "msm_internal.cuh"

namespace h2 {

std::atomic<double> g_lane_fraction{1.0};
std::atomic<size_t> g_pipe_chunk{0};   // h2_set_option("host_commit_chunk"): sweeps only

// Two-pass sort geometry for a table of `stride` columns and window width c: the low `lowb` bucket bits ride in the
// entry above the table index (`lb` bits); the remaining bucket_bits - lowb bits select the pass-1 bin.
bool sort2_geometry(u32 stride, int c, int *lowb_out, int *lb_out, int *side_out, u32 *s1_out) {
    const int W = 255 / c + 1, bucket_bits = c - 1;
    const uint64_t top = (uint64_t)W * stride - 1;
    if (top >= ((uint64_t)1 << 31)) return false;
    int lb = 0;
    while ((top >> lb) != 0) ++lb;
    // 512 pass-1 bins whenever the low bucket bits have somewhere to ride: in the entry's spare bits above the table index, or
    // -- windows of 18 bits and more at 2^20 points, where those are too few -- in a 16-bit SIDE array next to the tagged list
    // (pass 1 writes 6 bytes per entry instead of 4; without it c = 20 meant 4096 bins and 6-entry runs)
    int lowb = std::min(31 - lb, bucket_bits - 9), side = 0;
    static const bool side_ok = [] { const char *e = ab_env("H2_SORT_SIDE"); return !(e && atoi(e) == 0); }();      // A/B switch
    // ... only where the entry's own spare bits would leave more than 1024 bins: at 17-bit windows over 2^20 points (1024 bins
    // without it) the side array buys nothing and costs 50 % more tagged traffic (measured: 1027 against 1026-1038 M/s)
    if (side_ok && bucket_bits - lowb > 10 && bucket_bits - 9 <= 14 && W <= 64) {
        lowb = bucket_bits - 9;
        side = 1;
    }
    if (lowb < 1 || bucket_bits - lowb > 12) return false;
    const size_t nh = (size_t)1 << (bucket_bits - lowb);
    // pass-1 stage in LDS: 2048 scalars' digits per workgroup, 1024 where narrower windows mean more digits per scalar (13-bit tables:
    // 20 digits) -- only for callers that ask (s1_out); the others keep the fixed 2048 they were measured with
    static const u32 s1_env = [] { const char *e = ab_env("H2_S1_SCALARS"); int v = e ? atoi(e) : 0; return (u32)(v == 512 || v == 1024 || v == 2048 ? v : 0); }();   // sweeps only
    u32 s1 = s1_out && s1_env ? s1_env : kS1Scalars;
    if (s1_out && (nh * 3 + 1 + (size_t)s1 * W) * 4 > kLdsCap) s1 = 1024;
    if ((nh * 3 + 1 + (size_t)s1 * W) * 4 > kLdsCap) return false;
    if (s1_out) *s1_out = s1;
    *lowb_out = lowb;
    *lb_out = lb;
    if (side_out) *side_out = side;
    return true;
}
// The paired commit (h2_commit_pair_device): two bucket slices, key = side * NB + bucket.  Pass 1 stages s1 scalars' digits per
// workgroup in LDS: 2048 for 16 windows, 1024 when narrower windows (smaller tables: more digits per scalar) would not fit.
bool pair_geometry(size_t m, int c, u32 stride, int *lowb_out, int *lb_out, u32 *nh_out, u32 *s1_out) {
    if (c > kMaxC || c < 2 || m < 8192) return false;
    const int W = 255 / c + 1;
    const u32 tb = 2u << (c - 1);
    const uint64_t top = (uint64_t)W * stride - 1;
    if (top >= ((uint64_t)1 << 31)) return false;
    int lb = 0, kb = 0;
    while ((top >> lb) != 0) ++lb;
    while (((u64)(tb - 1) >> kb) != 0) ++kb;
    const int lowb = std::min(31 - lb, std::max(1, kb - 9));
    if (lowb < 1) return false;
    const u32 nh = (tb + (1u << lowb) - 1) >> lowb;
    if (nh > 4096) return false;
    u32 s1 = kS1Scalars;
    while (s1 >= 1024 && ((size_t)nh * 3 + 1 + (size_t)s1 * W) * 4 > kLdsCap) s1 /= 2;
    if (s1 < 1024) return false;
    *lowb_out = lowb;
    *lb_out = lb;
    *nh_out = nh;
    *s1_out = s1;
    return true;
}

// window width: minimise mixed adds + reduce work.  `shared_buckets`: registered bases (one slice).
// The generic path always splits scalars with the endomorphism (glv.cuh): the window Horner it halves (0.35 ms) and the
// smaller fold (9 slices instead of 16) outweigh the extra bucket additions (2n x 9 windows against n x 16, and the
// on-the-fly phi) at every size measured, 2.76 against 3.09 ms even at 2^20.  The size cap only keeps 2n below 2^31.

int choose_c(size_t n, bool shared_buckets) {
    auto feasible = [&](int c) {
        if (c <= kMaxC) return true;
        int lowb, lb;
        return shared_buckets && n + 1 < ((size_t)1 << 31) && sort2_geometry((u32)n + 1, c, &lowb, &lb);
    };
    if (const char *e = ab_env("H2_MSM_C")) {   // tuning sweeps only; the only way to windows beyond 16 bits (see below)
        int v = atoi(e);
        if (v >= 4 && v <= kMaxCShared && feasible(v)) return v;
    }
    // Windows of 17..20 bits (registered path) are implemented and parity-tested but not chosen: at n = 2^20, c = 20 cuts
    // the accumulate from 1.29 to 1.07 ms (13 windows instead of 16) and loses more than that in the sort (4096 pass-1
    // bins: 6-entry runs) and in the fold over 2^19 buckets (0.49 vs 0.27 ms).  DESIGN.md section 8.
    const bool glv = !shared_buckets && glv_applies(n);
    // With the split, magnitudes have 128 bits, so the TOP window of width c holds 128 - c floor(127 / c) significant
    // bits: 2 for c = 14, 8 for c = 12 or 15 -- every entry of that window then lands in a few hundred buckets, which are
    // summed by the (slow) heavy-bucket path.  c = 10, 13 and 16 fill their top window (8 of 10, 11 of 13, 16 of 16 bits).
    // Small multiexps are pure latency (a chain of ~128 doublings plus the per-slice folds); measured over c = 4..14
    // (bench/tools/c_sweep_small.py): 0.67-0.72 ms at c = 10 up to 2^12 points, c = 13 up to 2^19 (2^18: 1.08 against 1.20 ms at
    // c = 16, 2^19: 1.43 against 1.52; 2^20: equal, and the accumulate is shorter with 16), c = 16 beyond.
    // (round 3, with the fold on the carry-free layer, bench/tools/generic_ms.py: 2^12 at 13 / 10 bits 0.461 / 0.474 ms; 2^18 at 16 / 13
    // bits 0.846 / 0.851; 2^19 1.11 / 1.165: 10 bits up to 2^11, 13 up to 2^18, 16 beyond)
    if (glv) return n <= 2048 ? 10 : n <= 262144 ? 13 : 16;
    // Registered tables: a commit below ~2^17 points is a chain of latency-bound kernels, not bucket arithmetic, and the widths
    // whose top window is nearly empty (255 mod c small: 12, 14) send that window through the heavy-bucket path.  Measured, one
    // commit alone (bench/tools/c_sweep_registered.py): 8 bits up to 2^9 points (0.21-0.27 ms), 10 up to 2^10 (0.32), 13 up to
    // 2^13 (0.35-0.39; 12 bits at 2^12: 0.59; a paired commit at 2^13: 0.39 against 0.44 with 16), 16 from 2^14 on (0.41-0.49;
    // the cost model below picked 13-14 bits there: 2^15 0.48 -> 0.42).  Re-measured in round 3 with the fold on the carry-free
    // layer (bench/tools/lone_commit_ms.py, H2_MSM_C): 2^14 / 2^15 / 2^16 at 16 bits 0.24 / 0.27 / 0.27 ms, at 14 bits 0.32 / 0.37 /
    // 0.42, at 13 bits 0.37 / 0.54 / 0.59; 2^13 at 16 / 13 bits 0.244 / 0.294; 2^12 0.254 / 0.236; 2^11 0.263 / 0.222; 2^10 0.248 / 0.200
    // (10 bits: 0.258); 2^9 0.251 / 0.196 (8 bits: 0.256); 2^8 at 13 / 8 bits 0.204 / 0.196: 8 bits up to 2^8, 13 up to 2^12, 16 beyond.
    if (shared_buckets) return n <= 384 ? 8 : n <= 6144 ? 13 : 16;
    double best = 1e300;
    int bc = 4;
    for (int c = 4; c <= kMaxC; ++c) {
        // with the endomorphism split: 2n half-length scalars, 130 / c + 1 windows
        int W = glv ? 130 / c + 1 : 255 / c + 1;
        double pairs = glv ? 2.0 * (double)n : (double)n;
        double buckets = (double)(1u << (c - 1)) * (shared_buckets ? 1 : W);
        double cost = (double)W * pairs * 10.5 + buckets * 60.0;
        if (cost < best) { best = cost; bc = c; }
    }
    return bc;
}

// m: digit columns (generic path with the endomorphism split: 2 x scalars, `glv` picks the shorter window count)
MsmShape make_shape(size_t m, int c, bool shared_buckets, bool glv) {
    MsmShape s;
    s.c = c;
    s.W = glv ? 130 / c + 1 : 255 / c + 1;
    s.NB = 1u << (c - 1);
    s.slices = shared_buckets ? 1 : (u32)s.W;
    s.m = m;
    s.items = shared_buckets ? (size_t)s.W * m : m;
    u32 B = (u32)((s.items + 65535) / 65536);
    s.B = B < 1 ? 1 : B;
    s.chunk = (u32)((s.items + s.B - 1) / s.B);
    s.total_buckets = s.slices * s.NB;
    return s;
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
__global__ void k_points_sum(const u32 *__restrict__ pts, u32 count, u32 *__restrict__ out, bool in_mont = true, int out_kind = H2_OUT_JACOBIAN,
                             bool out_mont = true) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    xyzz<FB> r = xyzz_identity<FB>();
    for (u32 i = 0; i < count; ++i) {
        const u32 *p = pts + 24 * (size_t)i;
        fe Z = fe_load(p + 16);
        if (fe_is_zero(Z)) continue;
        xyzz<FB> q;
        q.x = fe_load(p);
        q.y = fe_load(p + 8);
        if (!in_mont) { q.x = fe_to_mont<FB>(q.x); q.y = fe_to_mont<FB>(q.y); Z = fe_to_mont<FB>(Z); }
        q.zz = fe_sqr<FB>(Z);
        q.zzz = fe_mulx<FB>(q.zz, Z);
        xyzz_add<FB>(r, q);
    }
    if (out_kind == H2_OUT_AFFINE) {
        affine<FB> a = xyzz_to_affine<FB>(r);
        if (!out_mont) { a.x = fe_from_mont<FB>(a.x); a.y = fe_from_mont<FB>(a.y); }
        fe_store(out, a.x);
        fe_store(out + 8, a.y);
        return;
    }
    fe X, Y, Z;
    xyzz_to_jacobian<FB>(r, X, Y, Z);
    if (!out_mont) { X = fe_from_mont<FB>(X); Y = fe_from_mont<FB>(Y); Z = fe_from_mont<FB>(Z); }
    fe_store(out, X);
    fe_store(out + 8, Y);
    fe_store(out + 16, Z);
}

// ---- debug timeline (H2_TIMELINE=1): one-lane stamp kernels between the stages of a commit record the device wall
// clock; h2_debug_timeline drains them.  Used to see how commits on different streams interleave on the chip.
__global__ void msm_stamp(unsigned long long *buf, u32 *count, u32 tag, u32 cap) {
    u32 i = atomicAdd(count, 1u);
    if (i < cap) {
        buf[2 * i] = wall_clock64();
        buf[2 * i + 1] = tag;
    }
}
static unsigned long long *g_tl_buf = nullptr;
static u32 *g_tl_count = nullptr;
static const u32 kTlCap = 1 << 16;
bool timeline_on() {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("H2_TIMELINE");
        on = e && atoi(e) ? 1 : 0;
        if (on) {
            if (hipMalloc(&g_tl_buf, kTlCap * 16) != hipSuccess || hipMalloc(&g_tl_count, 4) != hipSuccess) on = 0;
            else (void)hipMemset(g_tl_count, 0, 4);
        }
    }
    return on == 1;
}
#define TL_STAMP(tag) do { if (timeline_on()) hipLaunchKernelGGL(msm_stamp, dim3(1), dim3(1), 0, st, g_tl_buf, g_tl_count, (u32)(tag), kTlCap); } while (0)

// One workspace per (device, stream): calls enqueued on different streams never share scratch.
static std::mutex g_ctx_mu;
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<MsmContext>> g_ctxs;
MsmContext &msm_ctx(hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &slot = g_ctxs[std::make_pair(dev, st)];
    if (!slot) slot.reset(new MsmContext());
    return *slot;
}
// Is a grouped generic multiexp (msm_generic.hip) of ANOTHER stream of this device still in flight?  hipEventQuery on the event each context
// records behind its last one: a few microseconds, no synchronisation.
bool msm_other_generic_in_flight(const MsmContext *self) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto &kv : g_ctxs) {
        MsmContext *c = kv.second.get();
        if (kv.first.first != dev || c == self || !c->gpending.load()) continue;
        if (c->gdone && hipEventQuery(c->gdone) == hipErrorNotReady) return true;
        c->gpending = false;
    }
    return false;
}
// h2_trim: the per-(device, stream) scratch of this device goes back to the allocator (the device is idle by then)
void msm_release_workspaces() {
    msm_release_host_pipe();
    msm_release_host_msm_pipe();
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto &kv : g_ctxs) {
        if (kv.first.first != dev) continue;
        std::lock_guard<std::mutex> cl(kv.second->mu);
        kv.second->release_all();
    }
}

template <int FB, int FS> static int msm_launch(MsmContext &cx, const MsmArgs &a, hipStream_t st) {
    size_t m = a.n_used + (a.d_extra_scalar ? 1 : 0);
    int rc;
    const bool fold_only = a.fold_from != nullptr;
    if (m == 0 && a.add_into) return H2_OK;          // an empty range adds nothing
    if (m == 0 && !fold_only) {
        if ((rc = cx.ssums.reserve(128)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.ssums.ptr, 0, 128, st));
        hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, cx.ssums.as<u32>(), 1, 0, (u32 *)a.d_out, a.out_kind,
                           a.form == H2_FORM_MONTGOMERY);
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    // generic path: every scalar is split k = k1 + k2 lambda (glv.cuh) into two half-length digit columns, i for P_i and
    // n + i for phi(P_i): as many bucket additions (2n x 9 windows against n x 16), half the doublings in the final Horner
    const bool glv = !a.table && !a.d_extra_scalar && glv_applies(a.n_used);
    const size_t scalars_n = m;
    if (glv) m *= 2;
    if (fold_only && m < 1) m = 1;                   // only the bucket geometry (c) matters to the fold
    MsmShape sh = make_shape(m, a.c, a.table, glv);
    const u32 K = a.ncols > 1 ? (u32)a.ncols : 1u;
    if (K > 1 && (K > (u32)kMaxCols || !a.table || a.pair_shift >= 0 || a.add_into || fold_only || !a.col_scalars || !a.col_outs || a.n_used == 0))
        return H2_ERR_BATCH_SHAPE;
    const bool pair = a.table && a.pair_shift >= 0;
    if (pair) {                       // one bucket slice per output
        sh.slices = 2;
        sh.total_buckets = 2 * sh.NB;
    }
    const u32 tb = sh.total_buckets, segs = tb / kSeg;
    const size_t all_items = (size_t)sh.W * m;
    if (all_items >= ((size_t)1 << 31)) return H2_ERR_ARGS;  // entry = table index | sign << 31
    const u32 nblocks = (tb + kScanBlock - 1) / kScanBlock;
    if (!cx.attr_set) {
        H2_HIP(hipFuncSetAttribute((const void *)msm_count, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        H2_HIP(hipFuncSetAttribute((const void *)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        cx.attr_set = true;
    }
    // registered tables are stored in M9 form (h2_bases_register); the generic path converts its bases per call (below)
    static const bool glv_on_fe9 = [] { const char *e = ab_env("H2_GENERIC_FE9"); return !(e && atoi(e) == 0); }();
    const bool m9 = (a.table && !glv) || (glv && glv_on_fe9);
    u32 &lanes = cx.lanes[FB][m9 ? 2 : glv ? 1 : 0];
    if (!lanes) {  // how many lanes of the accumulate kernel the chip holds at once
        int dev = 0, cus = 0, per_cu = 0;
        H2_HIP(hipGetDevice(&dev));
        H2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (m9) H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, false, true>, 256, 0));
        else if (glv) H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, true>, 256, 0));
        else if (false) H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, false, true>, 256, 0));
        else H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, false>, 256, 0));
        // the M9 accumulate is sized for H2_ACC9_WAVES workgroups per CU even where its register count would let a third one in:
        // the wave slots and registers left over are what the sort / fold kernels of commits on OTHER streams run in
        // (H2_ACC_WAVES: sweeps only)
        static const int acc_waves = [] { const char *e = ab_env("H2_ACC_WAVES"); int v = e ? atoi(e) : 0; return v >= 1 && v <= 4 ? v : H2_ACC9_WAVES; }();
        if (m9) per_cu = std::min(per_cu, acc_waves);
        lanes = (u32)cus * (u32)std::max(per_cu, 1) * 256u;
    }
    // one round of resident lanes; small problems use fewer lanes so a range keeps >= 16 entries
    // a lane fraction < 1 (h2_set_option) leaves wave slots free so that the latency-bound sort / reduce kernels
    // of a commit running on ANOTHER stream can overlap this kernel (independent column commits)
    const double fraction = a.lane_fraction > 0.0 ? a.lane_fraction : g_lane_fraction.load();
    // H2_ACC_OVERSUB = k (sweeps only): k times as many, k times shorter lanes than the chip holds at once -- workgroups then enter as
    // slots free up, which evens out a launch that found some CUs half taken by other streams' sort / fold kernels, at the price of
    // k times the range heads for the finisher
    static const u32 oversub = [] { const char *e = ab_env("H2_ACC_OVERSUB"); int v = e ? atoi(e) : 0; return (u32)(v >= 1 && v <= 8 ? v : 1); }();
    const u32 usable = std::max(256u, (u32)(lanes * fraction) / 256u * 256u) * (a.table ? oversub : 1u);
    // entries per lane of the accumulate: 16 for full-size columns; small commits are chains of latency-bound kernels and run
    // shorter with more, shorter lanes (one registered commit at 2^11 .. 2^15 points: 3-7 % faster at 8; H2_MSM_DIV: sweeps only)
    static const u32 env_div = [] { const char *e = ab_env("H2_MSM_DIV"); int v = e ? atoi(e) : 0; return (u32)(v >= 1 && v <= 64 ? v : 0); }();
    const u32 lane_div = env_div ? env_div : (all_items < ((size_t)1 << 20) ? 8u : 16u);
    // Column-batched commits are JOINED (ColStride::joined) unless H2_BATCH_JOIN=0: the K sorted lists form one, which ONE launch of
    // msm_accumulate cuts into equal ranges -- the chip is tiled exactly as by a single commit (a launch per column leaves its last
    // round of workgroups ragged, and K of them next to each other share CUs unevenly), and the finisher meets T range heads per
    // batch instead of per column.
    static const bool join_env = [] { const char *e = ab_env("H2_BATCH_JOIN"); return !(e && e[0] == '0'); }();
    const bool joined = K > 1 && join_env;
    u32 T = (u32)std::min<size_t>(usable, std::max<size_t>(256, ((joined ? K : 1) * all_items / lane_div + 255) / 256 * 256));
    size_t head_slots = joined ? (size_t)T : (size_t)T * K;            // range heads parked in cx.seg9, in front of the K x tb bucket slots
    const u32 max_heavy = kMaxHeavy;
    // two-pass sort (registered path): always for windows beyond 16 bits, else for large bucket counts
    Sort2 S2;
    memset(&S2, 0, sizeof S2);
    S2.pair_shift = -1;
    static const u32 run_lanes_env = [] { const char *e = ab_env("H2_S1_RUN_LANES"); int v = e ? atoi(e) : 0; return (u32)(v == 8 || v == 16 || v == 32 || v == 64 ? v : 16); }();
    S2.run_lanes = run_lanes_env;
    bool use_sort2 = false;
    if (pair) {
        // key = side * NB + bucket over both slices (the generic path's multi-slice geometry), entry = table index
        int lb = 0, lowb = 0;
        u32 nh = 0, s1 = 0;
        if (!pair_geometry(m, sh.c, a.stride, &lowb, &lb, &nh, &s1)) return H2_ERR_ARGS;
        use_sort2 = true;
        S2.m = (u32)m; S2.c = sh.c; S2.W = sh.W; S2.mont = a.form == H2_FORM_MONTGOMERY;
        S2.stride = a.stride; S2.extra_col = 0xFFFFFFFFu;
        S2.lowb = lowb; S2.lb = lb; S2.nh = nh;
        S2.s1_scalars = s1;
        S2.nb = sh.NB;
        S2.B1 = (u32)((m + s1 - 1) / s1);
        S2.K2 = kS2Chunk;
        S2.B2 = (u32)((all_items + kS2Chunk - 1) / kS2Chunk);
        S2.lds_window = std::min<u32>(tb, 32768u);
        S2.pair_shift = a.pair_shift;
        S2.pair_n = a.pair_n;
    } else if (a.table && (sh.c > kMaxC || (sh.NB >= 4096 && (m >= 8192 || K > 1)))) {
        // (a column-batched commit exists in the two-pass form only, so it takes it from 13-bit tables on whatever the column length:
        // eight 2^12-point columns in one launch set are 0.3 ms against 0.9 ms for eight chains of one-pass sorts)
        static const int force_old = [] { const char *e = ab_env("H2_MSM_SORT"); return e && atoi(e) == 1 ? 1 : 0; }();
        int lowb = 0, lb = 0, side = 0;
        u32 s1 = kS1Scalars;
        if (sort2_geometry(a.stride, sh.c, &lowb, &lb, &side, &s1) && (sh.c > kMaxC || !force_old)) {
            use_sort2 = true;
            S2.m = (u32)m; S2.c = sh.c; S2.W = sh.W; S2.mont = a.form == H2_FORM_MONTGOMERY;
            S2.stride = a.stride; S2.extra_col = a.d_extra_scalar ? a.extra_col : 0xFFFFFFFFu;
            S2.lowb = lowb; S2.lb = lb; S2.nh = sh.NB >> lowb;
            S2.side = side;
            S2.s1_scalars = s1;
            S2.nb = sh.NB;
            S2.B1 = (u32)((m + s1 - 1) / s1);
            S2.K2 = kS2Chunk;
            S2.B2 = (u32)((all_items + kS2Chunk - 1) / kS2Chunk);
            S2.lds_window = std::min<u32>(sh.NB, 32768u);
            S2.col0 = a.col0;
        }
    } else if (glv && scalars_n >= 65536) {
        // generic path, large: sort key = window * NB + bucket over all slices, entry = digit column (< 2 * scalars)
        static const int force_old = [] { const char *e = ab_env("H2_MSM_SORT"); return e && atoi(e) == 1 ? 1 : 0; }();
        int lb = 0, kb = 0;
        while (((u64)(m - 1) >> lb) != 0) ++lb;
        while (((u64)(tb - 1) >> kb) != 0) ++kb;
        // bins of ~16 K entries (kb - 11 bucket bits per bin: 1152 bins for 9 slices of 2^15 buckets), so that pass 2 is the
        // one-launch form with a bin per workgroup in LDS; H2_GLV_BIN_BITS: sweeps only (9 = the chunked pass 2 of round 2)
        // Up to 2^19 scalars only: the carry slice of the split (the window above the top of a 128-bit half) puts ~n / 2 entries
        // into ONE bucket, and a bin that large was scattered by a single workgroup (2^19: sort 0.28 -> 0.16 ms; 2^20: 0.30 -> 0.61).
        // With the oversized-bin kernels (msm_s2_big_*) that bin is chunked over 64 workgroups: 2^20 takes the one-launch form with
        // 10 bits (1.83 -> 1.71 ms on one box; 11 bits 1.80, 12 bits 1.83); from 2^21 the forms are equal within 1 %.
        static const int glv_bin_bits = [] { const char *e = ab_env("H2_GLV_BIN_BITS"); int v = e ? atoi(e) : 0; return v >= 8 && v <= 12 ? v : 0; }();
        const int bin_bits = glv_bin_bits ? glv_bin_bits : (scalars_n <= ((size_t)1 << 19) ? 11 : scalars_n <= ((size_t)1 << 20) ? 10 : 9);
        const int lowb = std::min(31 - lb, std::max(1, kb - bin_bits));
        const u32 nh = (tb + (1u << lowb) - 1) >> lowb;
        const u32 s1 = 1024;
        const bool fits = ((size_t)nh * 3 + 1 + (size_t)s1 * 2 * sh.W) * 4 <= kLdsCap;
        if (!force_old && lowb >= 1 && nh <= 4096 && fits) {
            use_sort2 = true;
            S2.m = (u32)scalars_n; S2.c = sh.c; S2.W = sh.W; S2.mont = a.form == H2_FORM_MONTGOMERY;
            S2.stride = 0; S2.extra_col = 0xFFFFFFFFu;
            S2.lowb = lowb; S2.lb = lb; S2.nh = nh;
            S2.s1_scalars = s1;
            S2.nb = sh.NB;
            S2.B1 = (u32)((scalars_n + s1 - 1) / s1);
            S2.K2 = kS2Chunk;
            S2.B2 = (u32)((all_items + kS2Chunk - 1) / kS2Chunk);
            S2.lds_window = std::min<u32>(tb, 32768u);
        }
    }
    if (sh.c > kMaxC && !use_sort2) return H2_ERR_ARGS;   // choose_c only picks wide windows the two-pass sort can take
    if (K > 1 && !use_sort2) return H2_ERR_BATCH_SHAPE;
    const bool wide_reduce = sh.NB > 32768u;              // implies the registered path (one slice)
    static const bool fold9_on = [] { const char *e = ab_env("H2_FOLD9"); return !(e && atoi(e) == 0); }();     // A/B switch
    // the fold on the carry-free layer (fold9_* kernels: registered tables from 16-bit windows, paired commits, and the window
    // slices of a large generic multiexp); a range of a chunked commit hands finished buckets on in the reference's form
    // (add_into), so it keeps the 8 x 32 finisher
    static const u32 fold9_min_nb = [] { const char *e = ab_env("H2_FOLD9_MIN_NB"); int v = e ? atoi(e) : 0; return (u32)(v >= 64 ? v : 128); }();
    const bool fold9 = fold9_on && sh.NB >= fold9_min_nb && sh.slices <= 16 && m9 && !a.add_into && !fold_only;      // (16: arrival counters of fold9_planes)
    if (K > 1 && !fold9) return H2_ERR_BATCH_SHAPE;
    if (a.slice_sums_only && !(fold9 && glv)) return H2_ERR_BATCH_SHAPE;
    // Slice split (round 5; generic multiexps from 2^19 points): the sorted list is ordered by (slice, bucket), so the upper slices
    // [split_k, slices) and the lower ones [0, split_k) are two contiguous halves of it.  They are accumulated one after the other on
    // `st`; as soon as the UPPER group is in its buckets its fold and its Horner chain -- (slices - 1) c ~ 128 dependent doublings, 0.25 ms
    // on one quad of lanes, which used to follow the whole accumulate -- run on a side stream beside the lower group's accumulate and
    // fold.  What is left behind the accumulate: the lower group's fold, (split_k - 1) c doublings and one addition.  The bases'
    // conversion to M9 form runs on the side stream beside the sort.  H2_GENERIC_SPLIT=0: off (A/B); = k: force the lower group's size.
    static const int split_env = [] { const char *e = ab_env("H2_GENERIC_SPLIT"); return e ? atoi(e) : -1; }();
    // The grouped form (round 6, msm_generic.hip; generic multiexps beyond 2^18 points): the endomorphism split once, the window slices
    // sorted / accumulated / folded in groups, upper slices first.  It answers H2_ERR_BATCH_SHAPE before enqueueing anything when the shape
    // is not its own; round 5's slice split below then still applies (and is the A/B arm, H2_GENERIC_GROUPED=0 in the laboratory build).
    if (glv && fold9 && m9 && a.phase == 0 && !a.slice_sums_only && K == 1 && !prof_enabled() && !timeline_on()) {
        rc = msm_generic_grouped<FB, FS>(cx, a, sh, scalars_n, lanes, st);
        if (rc != H2_ERR_BATCH_SHAPE) return rc;
    }
    int split_k = 0;
    if (glv && fold9 && m9 && a.phase == 0 && !a.slice_sums_only && K == 1 && sh.slices >= 6 && split_env != 0 && !prof_enabled() && !timeline_on()) {
        if (split_env > 0) split_k = std::min<int>(split_env, (int)sh.slices - 2);
        else if (scalars_n >= ((size_t)1 << 19)) split_k = 3;
    }
    if (split_k) head_slots = 2 * (size_t)T;               // each group's T range heads
    // pass 2 of the two-pass sort in its one-launch form (a workgroup per pass-1 bin)?  Decided here, before anything is launched,
    // because a column-batched commit exists in that form only.
    bool s2_bins_form = false;
    size_t s2_cap_entries = 0;
    if (use_sort2) {
        static const bool bins_on = [] { const char *e = ab_env("H2_S2_BINS"); return !(e && atoi(e) == 0); }();
        const size_t nbk = (size_t)1 << S2.lowb;
        // LDS stage: the average bin + 25 % (two workgroups per CU where that fits: 2^20 scalars at 17 bits, 15 K-entry bins), at
        // most what one workgroup can have; a bin beyond its stage takes the direct-scatter branch.  H2_S2_CAP: sweeps only.
        const size_t cap_max = nbk * 8 + 64 < kLdsCap ? (kLdsCap - nbk * 8) / 4 : 0;
        static const size_t cap_env = [] { const char *e = ab_env("H2_S2_CAP"); return e ? (size_t)atol(e) : (size_t)0; }();
        s2_cap_entries = std::min(cap_max, cap_env ? cap_env : std::max<size_t>(4096, all_items / S2.nh * 5 / 4 + 1024));
        const size_t nbins = ((size_t)tb + nbk - 1) >> S2.lowb;
        s2_bins_form = bins_on && S2.lowb <= 12 && nbins == S2.nh && s2_cap_entries && all_items / S2.nh <= s2_cap_entries * 9 / 10;
    }
    if (K > 1 && !s2_bins_form) return H2_ERR_BATCH_SHAPE;
    u32 wideS = 0, wideNR = 0;
    if (wide_reduce || fold9) {
        const int bb = sh.c - 1;
        wideS = 1u << (bb / 2);
        wideNR = sh.NB / wideS;
    }
    size_t plan_words = 0;
    if (use_sort2) {
        if (!cx.attr2_set) {
            H2_HIP(hipFuncSetAttribute((const void *)msm_s2_count, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            H2_HIP(hipFuncSetAttribute((const void *)msm_s2_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            H2_HIP(hipFuncSetAttribute((const void *)msm_s1_scatter<FP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            H2_HIP(hipFuncSetAttribute((const void *)msm_s1_scatter<FQ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            H2_HIP(hipFuncSetAttribute((const void *)msm_s1_scatter<FP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            H2_HIP(hipFuncSetAttribute((const void *)msm_s1_scatter<FQ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            cx.attr2_set = true;
        }
        if ((rc = cx.hist.reserve((size_t)K * S2.B1 * S2.nh * 4)) != H2_OK) return rc;
        if ((rc = cx.tagged.reserve((size_t)K * all_items * 4)) != H2_OK) return rc;
        if (S2.side && (rc = cx.tagged_low.reserve((size_t)K * all_items * 2 + 64)) != H2_OK) return rc;
        plan_words = (size_t)S2.nh * 2 + 1 + (size_t)S2.B2 * 2 + 1 +
                     std::max<size_t>(((size_t)S2.nh + S2.B2 + 1) << S2.lowb, 64 + (((size_t)kMaxBig * (kBigChunks + 1)) << S2.lowb));
        plan_words = (plan_words + 3) & ~(size_t)3;
        if ((rc = cx.plan.reserve((size_t)K * plan_words * 4)) != H2_OK) return rc;
    } else {
        if ((rc = cx.digits.reserve(all_items * 2)) != H2_OK) return rc;
        if ((rc = cx.hist.reserve((size_t)sh.slices * sh.B * sh.NB * 4)) != H2_OK) return rc;
    }
    if ((rc = cx.counts.reserve((size_t)tb * 4)) != H2_OK) return rc;
    if ((rc = cx.starts.reserve((size_t)K * (tb + 2) * 4)) != H2_OK) return rc;
    if ((rc = cx.bsums.reserve((size_t)(nblocks + 4) * 4)) != H2_OK) return rc;
    if ((rc = cx.entries.reserve((size_t)K * all_items * 4)) != H2_OK) return rc;
    if ((rc = cx.heads.reserve((size_t)std::max<size_t>(T, (size_t)sh.slices * 32) * 128)) != H2_OK) return rc;
    if ((rc = cx.heavy.reserve((size_t)(split_k ? 2 : K) * (max_heavy + 2) * 4)) != H2_OK) return rc;
    if ((rc = cx.hscratch.reserve((size_t)(split_k ? 2 : K) * max_heavy * kHeavyBlocks * 144)) != H2_OK) return rc;
    if ((rc = cx.buckets.reserve((size_t)tb * 128)) != H2_OK) return rc;
    if ((rc = cx.partial.reserve(std::max(wide_reduce ? ((size_t)2 * wideNR / kSeg + 2 * wideNR) * 128 : (size_t)segs * 128,
                                          fold9 ? (size_t)K * sh.slices * (wideS + wideNR + 32) * 144 : (size_t)0))) != H2_OK) return rc;
    if (fold9 && cx.fold_ctr.cap < (size_t)K * 64) {      // fold9_planes' arrival counters (16 words per column): zero once, every launch leaves them at zero
        if ((rc = cx.fold_ctr.reserve((size_t)kMaxCols * 64)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.fold_ctr.ptr, 0, (size_t)kMaxCols * 64, st));
    }
    // column-batched commit: the per-column pointers and the distances between the per-column work areas (32-bit words)
    ColIn ci;
    ColOut co;
    ColStride cs;
    memset(&ci, 0, sizeof ci);
    memset(&co, 0, sizeof co);
    memset(&cs, 0, sizeof cs);
    if (K > 1) {
        for (u32 k = 0; k < K; ++k) {
            ci.scalars[k] = (const u32 *)a.col_scalars[k];
            ci.blinds[k] = a.col_blinds ? (const u32 *)a.col_blinds[k] : nullptr;
            co.out[k] = (u32 *)a.col_outs[k];
            if (!ci.scalars[k] || !co.out[k] || (a.d_extra_scalar && !ci.blinds[k])) return H2_ERR_ARGS;
        }
        cs.hist = (u32)((size_t)S2.B1 * S2.nh);
        cs.plan = (u32)plan_words;
        cs.items = (u32)all_items;
        cs.entries = joined ? 0u : (u32)all_items;
        cs.joined = joined ? S2.nh : 0u;
        cs.starts = joined ? tb : tb + 2;
        cs.heavy = max_heavy + 2;
        cs.hscratch = max_heavy * kHeavyBlocks * 36;
        cs.heads = T * 36;
        cs.buckets = tb * 36;
        cs.lines = sh.slices * (wideS + wideNR) * 36;
        cs.planes = sh.slices * 32 * 36;
        cs.ctr = 16;
    }
    if ((rc = cx.ssums.reserve((size_t)(std::max<u32>(sh.slices, 2) + 1) * 128)) != H2_OK) return rc;      // (+ 1: the upper group's weighted sum of a slice split)
    if (split_k && !cx.side) {
        H2_HIP(hipStreamCreateWithFlags(&cx.side, hipStreamNonBlocking));
        for (hipEvent_t *e : {&cx.ev_fork, &cx.ev_conv, &cx.ev_acc_a, &cx.ev_join}) H2_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    const u32 m32 = (u32)m;
    u32 *grand = cx.bsums.as<u32>() + nblocks;
    const u32 tl_id = (u32)(((uintptr_t)st >> 4) & 0xFFFF) << 8;
    // the one-launch pass 2 also clears the raw bucket slots (its workgroups own disjoint bucket ranges), so the slots must exist
    // before the sort is enqueued; a reservation that grows frees and synchronises, which is harmless here, in front of everything
    const bool zero_in_sort = m9 && use_sort2 && s2_bins_form && !fold_only;
    if (zero_in_sort && (rc = cx.seg9.reserve((head_slots + (size_t)K * tb) * 144)) != H2_OK) return rc;
    // oversized pass-2 bins (degenerate columns) go to the chunked msm_s2_big_* kernels only where a bin can be large enough for
    // that to matter: below 3 * 2^20 entries per column (2^18 scalars) the bin's own workgroup streams it (<= 2^17 entries: tens of
    // microseconds, and only for such columns), and every commit saves three launches that would find an empty list
    const u32 max_big = all_items >= ((size_t)3 << 20) ? kMaxBig : 0u;
    if (!fold_only) {
    TL_STAMP(tl_id | 1);
    if (split_k) {          // the bases' conversion (it reads nothing the sort writes) on the side stream, beside the sort
        if ((rc = cx.bases9.reserve((size_t)scalars_n * 128 + 64)) != H2_OK) return rc;
        H2_HIP(hipEventRecord(cx.ev_fork, st));
        H2_HIP(hipStreamWaitEvent(cx.side, cx.ev_fork, 0));
        hipLaunchKernelGGL((msm_bases_to_m9_glv<FB>), dim3(((u32)scalars_n + 255) / 256), dim3(256), 0, cx.side, (const u32 *)a.d_bases,
                           cx.bases9.as<u32>(), (u32)scalars_n);
        H2_HIP(hipEventRecord(cx.ev_conv, cx.side));
    }
    if (a.phase < 2) prof_begin(PROF_MSM_SORT, st);          // (phases >= 2 resume behind a sort the phase-1 call enqueued and timed)
    const u32 extra_col = a.d_extra_scalar ? (a.table ? a.extra_col : (u32)a.n_used) : 0xFFFFFFFFu;
    if (a.phase >= 2) {
        // the sort was enqueued by the phase-1 call
    } else if (use_sort2) {
        u32 *hist1 = cx.hist.as<u32>(), *bin_count = cx.plan.as<u32>(), *bin_start = bin_count + S2.nh, *hlo = bin_start + S2.nh + 1,
            *woff = hlo + S2.B2, *hist2 = woff + S2.B2 + 1;
        const size_t lds1 = ((size_t)S2.nh * 3 + 1 + (size_t)S2.s1_scalars * (glv ? 2 : 1) * sh.W) * 4;
        // 512 lanes per pass-1 workgroup: msm_s1_scatter takes 72 registers a lane, and 16 waves of it do not fit beside the two
        // msm_accumulate waves a SIMD already holds (2 x 168 of 512 registers) -- with 1024 lanes the sort of the NEXT commit on
        // another stream sat out the whole accumulate (416 us on average in a 3-stream trace against 57 us alone); LDS is free
        // there, the accumulate uses none.  H2_S1_THREADS: sweeps only.
        static const u32 s1_threads = [] { const char *e = ab_env("H2_S1_THREADS"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : 512); }();
        if (glv) {
            hipLaunchKernelGGL((msm_s1_count<FS, true>), dim3(S2.B1), dim3(s1_threads), S2.nh * 4, st, (const u32 *)a.d_scalars,
                               (const u32 *)nullptr, S2, hist1, ci, cs);
            hipLaunchKernelGGL(msm_s1_prefix, dim3((S2.nh + 15) / 16), dim3(1024), 0, st, hist1, bin_count, S2.B1, S2.nh, cx.heavy.as<u32>(), hist2, cx.starts.as<u32>() + tb + 1, cs);
            hipLaunchKernelGGL((msm_s1_scatter<FS, true>), dim3(S2.B1), dim3(s1_threads), lds1, st, (const u32 *)a.d_scalars,
                               (const u32 *)nullptr, S2, hist1, bin_count, bin_start, cx.tagged.as<u32>(), (uint16_t *)nullptr, ci, cs);
        } else {
            hipLaunchKernelGGL((msm_s1_count<FS, false>), dim3(S2.B1, 1, K), dim3(s1_threads), S2.nh * 4, st, (const u32 *)a.d_scalars,
                               (const u32 *)a.d_extra_scalar, S2, hist1, ci, cs);
            ColStride cs1 = cs;                   // joined columns: one heavy-bucket list and one sentinel, behind the K x tb boundaries
            if (joined) cs1.heavy = cs1.starts = 0;
            hipLaunchKernelGGL(msm_s1_prefix, dim3((S2.nh + 15) / 16, 1, K), dim3(1024), 0, st, hist1, bin_count, S2.B1, S2.nh, cx.heavy.as<u32>(), hist2,
                               cx.starts.as<u32>() + (joined ? (size_t)K * tb : (size_t)tb) + 1, cs1);
            hipLaunchKernelGGL((msm_s1_scatter<FS, false>), dim3(S2.B1, 1, K), dim3(s1_threads), lds1, st, (const u32 *)a.d_scalars,
                               (const u32 *)a.d_extra_scalar, S2, hist1, bin_count, bin_start, cx.tagged.as<u32>(), cx.tagged_low.as<uint16_t>(), ci, cs);
        }
        // pass 2: one launch, a workgroup per bin, when an average bin fits LDS with room to spare (registered tables; the 9-slice
        // generic sort has bins of ~64 K entries and keeps the chunked form); H2_S2_BINS=0: the chunked form (A/B)
        const size_t nbk = (size_t)1 << S2.lowb;
        const size_t cap_entries = s2_cap_entries;
        if (s2_bins_form) {
            if (!cx.attr_bins_set) {                              // per (device, stream) context: the attribute is per device
                H2_HIP(hipFuncSetAttribute((const void *)msm_s2_bins, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
                cx.attr_bins_set = true;
            }
            u32 *big = hist2, *gcnt = hist2 + 64;                 // the chunked form's histogram area is free here; msm_s1_prefix zeroed *big
            // the oversized-bin kernels return at once when the list is empty (the common case).  256-lane workgroups: a 1024-lane
            // workgroup of an EMPTY launch still needs four wave slots on every SIMD of one CU, and sat behind other streams'
            // accumulate for 10-160 us (profiles/r03_kernel_stats_3streams.csv) before it could find out that it had nothing to do
            static const u32 big_threads = [] { const char *e = ab_env("H2_S2_BIG_THREADS"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : 256); }();
            hipLaunchKernelGGL(msm_s2_bins, dim3(S2.nh, 1, K), dim3(1024), (nbk * 2 + cap_entries) * 4, st, cx.tagged.as<u32>(),
                               (const uint16_t *)cx.tagged_low.as<uint16_t>(), bin_start, S2, tb, (u32)cap_entries, cx.starts.as<u32>(), cx.entries.as<u32>(), big, max_big,
                               zero_in_sort ? cx.seg9.as<u32>() + 36 * head_slots : (u32 *)nullptr, cs);
            if (max_big) {
            hipLaunchKernelGGL(msm_s2_big_count, dim3(kBigChunks, kMaxBig, K), dim3(big_threads), nbk * 4, st, cx.tagged.as<u32>(),
                               (const uint16_t *)cx.tagged_low.as<uint16_t>(), bin_start, S2, (const u32 *)big, gcnt, cs);
            hipLaunchKernelGGL(msm_s2_big_prefix, dim3(kMaxBig, 1, K), dim3(big_threads), nbk * 4, st, bin_start, S2, tb, (const u32 *)big, gcnt, cx.starts.as<u32>(), cs);
            hipLaunchKernelGGL(msm_s2_big_scatter, dim3(kBigChunks, kMaxBig, K), dim3(big_threads), nbk * 4, st, cx.tagged.as<u32>(),
                               (const uint16_t *)cx.tagged_low.as<uint16_t>(), bin_start, S2, (const u32 *)big, (const u32 *)gcnt, cx.entries.as<u32>(), cs);
            }
        } else {
        hipLaunchKernelGGL(msm_s2_plan, dim3(1), dim3(kScanBlock), 0, st, bin_start, S2, hlo, woff);
        const size_t hist2_words = ((size_t)S2.nh + S2.B2 + 1) << S2.lowb;
        if (tb > S2.lds_window) H2_HIP(hipMemsetAsync(hist2, 0, hist2_words * 4, st));   // the HBM-counted windows start from zero
        const size_t lds2 = ((size_t)S2.lds_window + S2.nh + 1) * 4;
        hipLaunchKernelGGL(msm_s2_count, dim3(S2.B2), dim3(1024), lds2, st, cx.tagged.as<u32>(), (const uint16_t *)cx.tagged_low.as<uint16_t>(), bin_start, hlo, woff, S2, hist2);
        hipLaunchKernelGGL(msm_s2_prefix, dim3((tb + 255) / 256), dim3(256), 0, st, hist2, bin_start, hlo, woff, S2, cx.counts.as<u32>(),
                           tb);
        hipLaunchKernelGGL(msm_scan_blocksums, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), tb);
        hipLaunchKernelGGL(msm_scan_top, dim3(1), dim3(kScanBlock), 0, st, cx.bsums.as<u32>(), nblocks, grand);
        hipLaunchKernelGGL(msm_scan_apply, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), grand,
                           cx.starts.as<u32>(), tb);
        const size_t lds2s = std::max<size_t>(lds2, ((size_t)kS2StageWindow * 3 + 1 + S2.nh + kS2Chunk) * 4 + (size_t)kS2Chunk * 2);
        hipLaunchKernelGGL(msm_s2_scatter, dim3(S2.B2), dim3(1024), lds2s, st, cx.tagged.as<u32>(), (const uint16_t *)cx.tagged_low.as<uint16_t>(), bin_start, hlo, woff, S2, hist2,
                           cx.starts.as<u32>(), cx.entries.as<u32>());
        }
    } else {
        if (glv)
            hipLaunchKernelGGL((msm_recode_glv<FS>), dim3(((u32)scalars_n + 255) / 256), dim3(256), 0, st, (const u32 *)a.d_scalars,
                               cx.digits.as<uint16_t>(), (u32)scalars_n, sh.c, sh.W, a.form == H2_FORM_MONTGOMERY);
        else
            hipLaunchKernelGGL((msm_recode<FS>), dim3((m32 + 255) / 256), dim3(256), 0, st, (const u32 *)a.d_scalars,
                               (const u32 *)a.d_extra_scalar, cx.digits.as<uint16_t>(), m32, sh.c, sh.W,
                               a.form == H2_FORM_MONTGOMERY);
        hipLaunchKernelGGL(msm_count, dim3(sh.B, sh.slices), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(),
                           cx.hist.as<u32>(), sh.items, sh.chunk, sh.NB);
        hipLaunchKernelGGL(msm_chunk_prefix, dim3((tb + 255) / 256), dim3(256), 0, st, cx.hist.as<u32>(), cx.counts.as<u32>(),
                           sh.NB, sh.B, tb);
        hipLaunchKernelGGL(msm_scan_blocksums, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), tb);
        hipLaunchKernelGGL(msm_scan_top, dim3(1), dim3(kScanBlock), 0, st, cx.bsums.as<u32>(), nblocks, grand);
        hipLaunchKernelGGL(msm_scan_apply, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), grand,
                           cx.starts.as<u32>(), tb);
        hipLaunchKernelGGL(msm_scatter, dim3(sh.B, sh.slices), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(),
                           cx.hist.as<u32>(), cx.starts.as<u32>(), cx.entries.as<u32>(), sh.items, sh.chunk, sh.NB, m32,
                           a.table ? a.stride : 0u, extra_col, a.table ? 1 : 0, a.table ? a.col0 : 0u);
    }
#ifdef H2_SORT_DEBUG
    if (use_sort2 && sh.c <= kMaxC && a.table) {
        std::vector<u32> sa(tb + 1), ea(all_items), sb(tb + 1), eb(all_items);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(sa.data(), cx.starts.ptr, (tb + 1) * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ea.data(), cx.entries.ptr, (size_t)sa[tb] * 4, hipMemcpyDeviceToHost);
        (void)cx.digits.reserve(all_items * 2);
        DevBuf h2b;
        (void)h2b.reserve((size_t)sh.slices * sh.B * sh.NB * 4);
        hipLaunchKernelGGL((msm_recode<FS>), dim3((m32 + 255) / 256), dim3(256), 0, st, (const u32 *)a.d_scalars,
                           (const u32 *)a.d_extra_scalar, cx.digits.as<uint16_t>(), m32, sh.c, sh.W, a.form == H2_FORM_MONTGOMERY);
        hipLaunchKernelGGL(msm_count, dim3(sh.B, sh.slices), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(), h2b.as<u32>(), sh.items, sh.chunk, sh.NB);
        hipLaunchKernelGGL(msm_chunk_prefix, dim3((tb + 255) / 256), dim3(256), 0, st, h2b.as<u32>(), cx.counts.as<u32>(), sh.NB, sh.B, tb);
        hipLaunchKernelGGL(msm_scan_blocksums, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), tb);
        hipLaunchKernelGGL(msm_scan_top, dim3(1), dim3(kScanBlock), 0, st, cx.bsums.as<u32>(), nblocks, grand);
        hipLaunchKernelGGL(msm_scan_apply, dim3(nblocks), dim3(kScanBlock), 0, st, cx.counts.as<u32>(), cx.bsums.as<u32>(), grand, cx.starts.as<u32>(), tb);
        hipLaunchKernelGGL(msm_scatter, dim3(sh.B, sh.slices), dim3(1024), sh.NB * 4, st, cx.digits.as<uint16_t>(), h2b.as<u32>(), cx.starts.as<u32>(),
                           cx.entries.as<u32>(), sh.items, sh.chunk, sh.NB, m32, a.table ? a.stride : 0u, extra_col, a.table ? 1 : 0, a.table ? a.col0 : 0u);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(sb.data(), cx.starts.ptr, (tb + 1) * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(eb.data(), cx.entries.ptr, (size_t)sb[tb] * 4, hipMemcpyDeviceToHost);
        h2b.release();
        size_t bad_s = 0, bad_e = 0, first_s = (size_t)-1, first_e = (size_t)-1;
        for (u32 j = 0; j <= tb; ++j) if (sa[j] != sb[j]) { if (!bad_s) first_s = j; ++bad_s; }
        if (!bad_s)
            for (u32 j = 0; j < tb; ++j) {
                std::sort(ea.begin() + sa[j], ea.begin() + sa[j + 1]);
                std::sort(eb.begin() + sb[j], eb.begin() + sb[j + 1]);
                if (!std::equal(ea.begin() + sa[j], ea.begin() + sa[j + 1], eb.begin() + sb[j])) { if (!bad_e) first_e = j; ++bad_e; }
            }
        fprintf(stderr, "[sort-debug] m=%zu NB=%u nh=%u lowb=%d lb=%d M=%u/%u starts mismatches %zu (first %zu: %u vs %u) bucket-content mismatches %zu (first %zu)\n",
                m, sh.NB, S2.nh, S2.lowb, S2.lb, sa[tb], sb[tb], bad_s, first_s, first_s != (size_t)-1 ? sa[first_s] : 0,
                first_s != (size_t)-1 ? sb[first_s] : 0, bad_e, first_e);
    }
#endif
    if (a.phase == 1) {
        // every workspace the rest needs is reserved NOW: a reservation that grows frees and synchronises, which the phase-2 call must
        // not do under the sort's feet
        if (m9) {
            if (glv && (rc = cx.bases9.reserve((size_t)scalars_n * 128 + 64)) != H2_OK) return rc;
            if ((rc = cx.seg9.reserve((head_slots + (size_t)K * tb) * 144)) != H2_OK) return rc;
        }
        prof_end(PROF_MSM_SORT, st);
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    if (split_k) {
        if ((rc = cx.seg9.reserve((head_slots + (size_t)tb) * 144)) != H2_OK) return rc;
        u32 *heads_b = cx.seg9.as<u32>(), *heads_a = heads_b + 36 * (size_t)T, *buckets9 = cx.seg9.as<u32>() + 36 * head_slots;
        if (!zero_in_sort) H2_HIP(hipMemsetAsync(buckets9, 0, (size_t)tb * 144, st));
        const u32 kb = (u32)split_k * sh.NB, tb_a = tb - kb, ns_a = sh.slices - (u32)split_k;       // buckets of the lower group; buckets / slices of the upper one
        u32 *heavy_b = cx.heavy.as<u32>(), *heavy_a = heavy_b + (max_heavy + 2);
        u32 *hscr_b = cx.hscratch.as<u32>(), *hscr_a = hscr_b + (size_t)max_heavy * kHeavyBlocks * 36;
        if (!use_sort2) H2_HIP(hipMemsetAsync(heavy_b, 0, 8, st));
        H2_HIP(hipMemsetAsync(heavy_a, 0, 8, st));
        H2_HIP(hipStreamWaitEvent(st, cx.ev_conv, 0));
        const u32 *pts = cx.bases9.as<u32>(), *starts = cx.starts.as<u32>();
        u32 *lines9 = cx.partial.as<u32>(), *planes9 = lines9 + 36 * (size_t)sh.slices * (wideS + wideNR), *ssums = cx.ssums.as<u32>();
        int cb = 0;
        while ((1u << cb) < wideS) ++cb;
        const bool mont = a.form == H2_FORM_MONTGOMERY;
        // the upper slices first, then the lower ones
        hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(T / 256), dim3(256), 0, st, pts, (const u32 *)nullptr, 0xFFFFFFFFu, cx.entries.as<u32>(),
                           starts + kb, heads_a, buckets9 + 36 * (size_t)kb, tb_a, T, lane_div, cs);
        H2_HIP(hipEventRecord(cx.ev_acc_a, st));
        hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(T / 256), dim3(256), 0, st, pts, (const u32 *)nullptr, 0xFFFFFFFFu, cx.entries.as<u32>(),
                           starts, heads_b, buckets9, kb, T, lane_div, cs);
        // a group's fold down to its slice sums: finish (range heads into their buckets), the heavy buckets, line sums, planes
        auto fold_group = [&](hipStream_t s_, const u32 *heads9, const u32 *gstarts, u32 *gbuckets, u32 *heavy, u32 *hscr, u32 gtb, u32 slice0, u32 nslices) {
            hipLaunchKernelGGL((fold9_finish<FB>), dim3((gtb + 255) / 256), dim3(256), 0, s_, heads9, gstarts, gbuckets, heavy, gtb, T, lane_div, cs);
            hipLaunchKernelGGL((fold9_finish_heavy<FB>), dim3(kHeavyBlocks, kHeavyRows), dim3(256), 0, s_, heads9, gstarts, hscr, (const u32 *)heavy, gtb, T, lane_div, cs);
            hipLaunchKernelGGL((fold9_finish_heavy2<FB>), dim3(kHeavyRows), dim3(64), 0, s_, (const u32 *)hscr, gbuckets, (const u32 *)heavy, cs);
            hipLaunchKernelGGL((fold9_rowcol<FB>), dim3(wideS + wideNR - 1, nslices), dim3(256), 0, s_, (const u32 *)gbuckets, lines9 + 36 * (size_t)slice0 * (wideS + wideNR),
                               wideS, wideNR, cs);
            hipLaunchKernelGGL((fold9_planes<FB>), dim3(sh.c - 1, nslices), dim3(256), 0, s_, (const u32 *)(lines9 + 36 * (size_t)slice0 * (wideS + wideNR)),
                               planes9 + 36 * (size_t)slice0 * 32, cx.fold_ctr.as<u32>() + slice0, wideS, wideNR, cb, ssums + 32 * (size_t)slice0, kOutSliceSum, mont, co, cs);
        };
        // upper group on the side stream: fold, Horner over its slices, split_k c more doublings -> one weighted point behind the slice sums
        H2_HIP(hipStreamWaitEvent(cx.side, cx.ev_acc_a, 0));
        fold_group(cx.side, heads_a, starts + kb, buckets9 + 36 * (size_t)kb, heavy_a, hscr_a, tb_a, (u32)split_k, ns_a);
        hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, cx.side, (const u32 *)(ssums + 32 * (size_t)split_k), (int)ns_a, sh.c, ssums + 32 * (size_t)sh.slices,
                           kOutSliceSum, 1, split_k * sh.c, (const u32 *)nullptr);
        H2_HIP(hipEventRecord(cx.ev_join, cx.side));
        // lower group behind its accumulate, then the two halves meet
        fold_group(st, heads_b, starts, buckets9, heavy_b, hscr_b, kb, 0u, (u32)split_k);
        H2_HIP(hipStreamWaitEvent(st, cx.ev_join, 0));
        hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, (const u32 *)ssums, split_k, sh.c, (u32 *)a.d_out, a.out_kind, mont ? 1 : 0, 0,
                           (const u32 *)(ssums + 32 * (size_t)sh.slices));
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    if (a.phase != 4) {              // (phase 4: the accumulate ran in a phase-3 call)
    if (m9) {
        if (glv && (rc = cx.bases9.reserve((size_t)scalars_n * 128 + 64)) != H2_OK) return rc;
        // raw M9 segments: the heads of the T ranges of every column, then the bucket slots of every column (zeroed in one go)
        if ((rc = cx.seg9.reserve((head_slots + (size_t)K * tb) * 144)) != H2_OK) return rc;
        if (!zero_in_sort) H2_HIP(hipMemsetAsync(cx.seg9.as<u32>() + 36 * head_slots, 0, (size_t)K * tb * 144, st));
    } else {
        H2_HIP(hipMemsetAsync(cx.buckets.ptr, 0, (size_t)tb * 128, st));
    }
    if (!use_sort2) H2_HIP(hipMemsetAsync(cx.heavy.ptr, 0, 8, st));          // (the two-pass sort's msm_s1_prefix zeroed it)
    if (a.phase < 2) prof_end(PROF_MSM_SORT, st);
    TL_STAMP(tl_id | 2);
    prof_begin(PROF_MSM_ACCUMULATE, st);
    if (glv && !m9)
        hipLaunchKernelGGL((msm_accumulate<FB, true>), dim3(T / 256), dim3(256), 0, st, (const u32 *)a.d_bases, (const u32 *)nullptr,
                           (u32)scalars_n, cx.entries.as<u32>(), cx.starts.as<u32>(), cx.heads.as<u32>(), cx.buckets.as<u32>(), tb, T, lane_div, cs);
    else if (m9) {
        const u32 *pts = (const u32 *)a.d_bases;
        if (glv) {
            hipLaunchKernelGGL((msm_bases_to_m9_glv<FB>), dim3(((u32)scalars_n + 255) / 256), dim3(256), 0, st, (const u32 *)a.d_bases,
                               cx.bases9.as<u32>(), (u32)scalars_n);
            pts = cx.bases9.as<u32>();
        }
        hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(T / 256, 1, joined ? 1 : K), dim3(256), 0, st, pts,
                           (const u32 *)nullptr, 0xFFFFFFFFu, cx.entries.as<u32>(), cx.starts.as<u32>(), cx.seg9.as<u32>(),
                           cx.seg9.as<u32>() + 36 * head_slots, joined ? K * tb : tb, T, lane_div, cs);
        if (!fold9)
            hipLaunchKernelGGL((msm_segments_to_r256<FB>), dim3((T + tb + 255) / 256), dim3(256), 0, st, cx.seg9.as<u32>(),
                               cx.heads.as<u32>(), cx.buckets.as<u32>(), T, tb);
    }
    else
        hipLaunchKernelGGL((msm_accumulate<FB, false>), dim3(T / 256), dim3(256), 0, st, (const u32 *)a.d_bases,
                           (const u32 *)a.d_extra_base, (!a.table && a.d_extra_base) ? (u32)a.n_used : 0xFFFFFFFFu,
                           cx.entries.as<u32>(), cx.starts.as<u32>(), cx.heads.as<u32>(), cx.buckets.as<u32>(), tb, T, lane_div, cs);
    prof_end(PROF_MSM_ACCUMULATE, st);
    TL_STAMP(tl_id | 3);
    if (a.phase == 3) {
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    }
    prof_begin(PROF_MSM_REDUCE, st);
    if (fold9) {
        // wide slice: finish on the raw M9 segments, one lane per bucket (fold9_* above); the buckets stay in cx.seg9
        u32 *heads9 = cx.seg9.as<u32>(), *buckets9 = cx.seg9.as<u32>() + 36 * head_slots;
        const u32 fz = joined ? 1 : K, ftb = joined ? K * tb : tb;       // joined columns: one pass over the K x tb buckets
        hipLaunchKernelGGL((fold9_finish<FB>), dim3((ftb + 255) / 256, 1, fz), dim3(256), 0, st, (const u32 *)heads9, cx.starts.as<u32>(), buckets9,
                           cx.heavy.as<u32>(), ftb, T, lane_div, cs);
        hipLaunchKernelGGL((fold9_finish_heavy<FB>), dim3(kHeavyBlocks, kHeavyRows, fz), dim3(256), 0, st, (const u32 *)heads9, cx.starts.as<u32>(),
                           cx.hscratch.as<u32>(), cx.heavy.as<u32>(), ftb, T, lane_div, cs);
        hipLaunchKernelGGL((fold9_finish_heavy2<FB>), dim3(kHeavyRows, 1, fz), dim3(64), 0, st, cx.hscratch.as<u32>(), buckets9, cx.heavy.as<u32>(), cs);
    } else {
    hipLaunchKernelGGL((msm_finish_buckets<FB>), dim3((tb * kGroup + 255) / 256), dim3(256), 0, st, cx.heads.as<u32>(),
                       cx.starts.as<u32>(), cx.buckets.as<u32>(), cx.heavy.as<u32>(), tb, T, lane_div);
    hipLaunchKernelGGL((msm_finish_heavy<FB>), dim3(kHeavyBlocks, max_heavy), dim3(256), (256 / kGroup) * 128, st,
                       cx.heads.as<u32>(), cx.starts.as<u32>(), cx.hscratch.as<u32>(), cx.heavy.as<u32>(), tb, T, lane_div);
    hipLaunchKernelGGL((msm_finish_heavy2<FB>), dim3(max_heavy), dim3(64), 0, st, cx.hscratch.as<u32>(), cx.buckets.as<u32>(),
                       cx.heavy.as<u32>());
    }
    if (a.add_into) {
        hipLaunchKernelGGL((msm_bucket_add<FB>), dim3((tb * kGroup + 255) / 256), dim3(256), 0, st, a.add_into, cx.buckets.as<u32>(), tb);
        prof_end(PROF_MSM_REDUCE, st);
        H2_HIP(hipGetLastError());
        return H2_OK;
    }
    } else {
        prof_begin(PROF_MSM_REDUCE, st);
    }
    {
        // what the fold runs over: the bucket slices themselves, or (wide slices) the row / column sums as two slices
        const u32 *fold_src = fold_only ? a.fold_from : cx.buckets.as<u32>();
        u32 fold_nb = sh.NB, fold_slices = sh.slices;
        int fold_c = sh.c;
        if (fold9) {
            // line sums, then the bit planes of the line weights and their combination in one launch (fold9_planes)
            u32 *lines9 = cx.partial.as<u32>(), *planes9 = lines9 + 36 * (size_t)K * sh.slices * (wideS + wideNR);
            int cb = 0;
            while ((1u << cb) < wideS) ++cb;
            hipLaunchKernelGGL((fold9_rowcol<FB>), dim3(wideS + wideNR - 1, sh.slices, K), dim3(256), 0, st,
                               (const u32 *)(cx.seg9.as<u32>() + 36 * head_slots), lines9, wideS, wideNR, cs);
            const bool windows = glv;                // the slices are window slices: their sums meet in msm_combine's Horner step
            hipLaunchKernelGGL((fold9_planes<FB>), dim3(sh.c - 1, sh.slices, K), dim3(256), 0, st, (const u32 *)lines9, planes9, cx.fold_ctr.as<u32>(),
                               wideS, wideNR, cb, windows ? cx.ssums.as<u32>() : (u32 *)a.d_out, windows ? kOutSliceSum : a.out_kind,
                               a.form == H2_FORM_MONTGOMERY, co, cs);
            if (windows && !a.slice_sums_only)
                hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, cx.ssums.as<u32>(), (int)sh.slices, sh.c, (u32 *)a.d_out, a.out_kind,
                                   a.form == H2_FORM_MONTGOMERY);
            prof_end(PROF_MSM_REDUCE, st);
            TL_STAMP(tl_id | 4);
            H2_HIP(hipGetLastError());
            return H2_OK;
        }
        if (wide_reduce) {
            u32 *wide = cx.partial.as<u32>() + 32 * (size_t)(2 * wideNR / kSeg);     // after the fold's own partials
            H2_HIP(hipMemsetAsync(wide, 0, (size_t)2 * wideNR * 128, st));
            hipLaunchKernelGGL((msm_rowcol_sums<FB>), dim3(wideS + wideNR - 1), dim3(256), (256 / kGroup) * 128, st, fold_src,
                               wide, wideS, wideNR);
            fold_src = wide;
            fold_nb = wideNR;
            fold_slices = 2;
            fold_c = (sh.c - 1) / 2;      // log2 S
        }
        // A segment is 2 seg running-sum additions + a small-scalar multiple (~23 more dependent operations) on one quad of lanes:
        // 4 buckets while the segments fit the chip a few times over (the fold is their latency: one commit 0.235 -> 0.218 ms,
        // small commits 5-8 %), 8 when there are many slices (generic multiexps of 2^20 points: the multiples are throughput)
        const int seg = (size_t)fold_slices * fold_nb / kSeg <= 32768 ? kSeg : 2 * kSeg;
        const u32 fold_segs = fold_slices * fold_nb / seg;
        hipLaunchKernelGGL((msm_reduce_segments<FB>), dim3((fold_segs * kGroup + 255) / 256), dim3(256), 0, st, fold_src,
                           cx.partial.as<u32>(), fold_nb, fold_segs, seg);
        // 64 logical lanes per workgroup; first level leaves <= 32 block sums per slice
        const u32 per_slice = fold_nb / seg, nl = 256 / kGroup;
        const u32 bps = std::max(1u, std::min(32u, per_slice / (2 * nl)));
        const u32 share = (per_slice + bps - 1) / bps;
        if (bps > 1) {
            hipLaunchKernelGGL((msm_sum_slice<FB>), dim3(bps, fold_slices), dim3(256), nl * 128, st, cx.partial.as<u32>(),
                               cx.heads.as<u32>(), per_slice, share);     // heads[] is free again: reuse as level-1 output
            hipLaunchKernelGGL((msm_sum_slice<FB>), dim3(1, fold_slices), dim3(256), nl * 128, st, cx.heads.as<u32>(),
                               cx.ssums.as<u32>(), bps, bps);
        } else {
            hipLaunchKernelGGL((msm_sum_slice<FB>), dim3(1, fold_slices), dim3(256), nl * 128, st, cx.partial.as<u32>(),
                               cx.ssums.as<u32>(), per_slice, per_slice);
        }
        hipLaunchKernelGGL((msm_combine<FB>), dim3(pair ? 2 : 1), dim3(64), 0, st, cx.ssums.as<u32>(), (int)fold_slices, fold_c, (u32 *)a.d_out,
                           a.out_kind, a.form == H2_FORM_MONTGOMERY);
    }
    prof_end(PROF_MSM_REDUCE, st);
    TL_STAMP(tl_id | 4);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int msm_dispatch(MsmContext &cx, int curve, const MsmArgs &a, hipStream_t st) {
    if (curve == H2_PALLAS) return msm_launch<FP, FQ>(cx, a, st);
    return msm_launch<FQ, FP>(cx, a, st);
}

void to_mont_async(int curve, u32 *d, size_t field_elems, hipStream_t st) {
    if (!field_elems) return;
    dim3 grid((unsigned)((field_elems + 255) / 256)), block(256);
    if (curve == H2_PALLAS) hipLaunchKernelGGL((k_to_mont<FP>), grid, block, 0, st, d, field_elems);
    else hipLaunchKernelGGL((k_to_mont<FQ>), grid, block, 0, st, d, field_elems);
}

}  // namespace h2

using namespace h2;

extern "C" int h2_msm_window_bits(size_t n) { return choose_c(n ? n : 1, false); }
extern "C" int h2_commit_window_bits(size_t n) { return choose_c(n ? n : 1, true); }
extern "C" int h2_commit_pair_supported(size_t n) {
    int lowb, lb;
    u32 nh, s1;
    return n >= 8 && n <= (1u << 26) && pair_geometry(n, choose_c(n, true), (u32)n + 1, &lowb, &lb, &nh, &s1) ? 1 : 0;
}

// copies up to `cap` {clock, tag} pairs recorded under H2_TIMELINE=1 (measurement aid, see the header)
extern "C" int h2_debug_timeline(unsigned long long *out, unsigned cap) {
    if (!timeline_on() || !out) return -1;
    u32 n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, g_tl_count, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    n = std::min(std::min(n, kTlCap), cap);
    if (hipMemcpy(out, g_tl_buf, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemset(g_tl_count, 0, 4);
    return (int)n;
}

extern "C" int h2_set_option(const char *key, double value) {
    if (!key) return H2_ERR_ARGS;
    if (strcmp(key, "msm_lane_fraction") == 0) {
        if (!(value > 0.05 && value <= 1.0)) return H2_ERR_ARGS;
        g_lane_fraction.store(value);
        return H2_OK;
    }
    if (strcmp(key, "host_commit_chunk") == 0) {        // scalars per range of a pipelined host commit (0 = default); sweeps only
        if (!(value >= 0 && value <= (double)(1u << 26))) return H2_ERR_ARGS;
        g_pipe_chunk.store((size_t)value);
        return H2_OK;
    }
    return H2_ERR_ARGS;
}

extern "C" int h2_msm_device(int curve, const void *d_scalars, const void *d_bases_xy, size_t n, int form, int out_kind,
                             void *d_out, void *stream) {
    if (bad_common(curve, form, out_kind) || !d_out || (n && (!d_scalars || !d_bases_xy)) || n > 0x7FFFFFF0u)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    const void *bases = d_bases_xy;
    if (form == H2_FORM_CANONICAL && n) {  // bases arrive canonical: convert a private copy to Montgomery
        if ((rc = cx.stage_b.reserve(n * 64)) != H2_OK) return rc;
        H2_HIP(hipMemcpyAsync(cx.stage_b.ptr, d_bases_xy, n * 64, hipMemcpyDeviceToDevice, st));
        to_mont_async(curve, cx.stage_b.as<u32>(), n * 2, st);
        bases = cx.stage_b.ptr;
    }
    MsmArgs a{d_scalars, nullptr, bases, nullptr, n, false, choose_c(n ? n : 1, false), 0, 0xFFFFFFFFu, form, out_kind, d_out};
    return msm_dispatch(cx, curve, a, st);
}

static int commit_device_impl(h2_bases_t g, const void *d_scalars, size_t n, const void *d_w_xy, const void *d_blind, int form,
                              int out_kind, void *d_out, void *stream, bool blind_base_ready, double lane_fraction = 0.0);

extern "C" int h2_commit_device(h2_bases_t g, const void *d_scalars, size_t n, const void *d_w_xy, const void *d_blind,
                                int form, int out_kind, void *d_out, void *stream) {
    return commit_device_impl(g, d_scalars, n, d_w_xy, d_blind, form, out_kind, d_out, stream, false);
}

static int commit_device_impl(h2_bases_t g, const void *d_scalars, size_t n, const void *d_w_xy, const void *d_blind, int form,
                              int out_kind, void *d_out, void *stream, bool blind_base_ready, double lane_fraction) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    // d_blind without d_w_xy: the handle's own blind base (h2_bases_set_blind_base).  d_w_xy without d_blind: nothing to multiply.
    if (bad_common(b->curve, form, out_kind) || !d_out || (n && !d_scalars) || n > b->n || (d_w_xy && !d_blind)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if (d_blind && !blind_base_ready) {
        if (d_w_xy) {
            if ((rc = override_blind_base_device(*b, d_w_xy, form, st)) != H2_OK) return rc;
        } else {
            std::lock_guard<std::mutex> bl(b->mu);
            if (!b->blind_set) {
                set_last_error_msg("commit with a blind but the handle has no blind base: call h2_bases_set_blind_base, or pass d_w_xy");
                return H2_ERR_ARGS;
            }
        }
    }
    MsmArgs a{d_scalars, d_blind, b->d_table, nullptr, n, true, b->c, b->stride, (u32)b->n, form, out_kind, d_out};
    a.lane_fraction = lane_fraction;
    return msm_dispatch(cx, b->curve, a, st);
}

// the commit restricted to table columns [first, first + n): d_scalars[i] multiplies registered base first + i.  One range of a
// commit that is split over GPUs (h2_commit_split_rccl_device) or over the chunks of a host transfer; the blind term (the
// handle's blind base) rides with whichever range passes d_blind.
extern "C" int h2_commit_range_device(h2_bases_t g, const void *d_scalars, size_t first, size_t n, const void *d_blind, int form,
                                      int out_kind, void *d_out, void *stream) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !d_out || (n && !d_scalars) || first > b->n || n > b->n - first) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (d_blind) {
        std::lock_guard<std::mutex> bl(b->mu);
        if (!b->blind_set) {
            set_last_error_msg("range commit with a blind but the handle has no blind base: call h2_bases_set_blind_base");
            return H2_ERR_ARGS;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    MsmArgs a{d_scalars, d_blind, b->d_table, nullptr, n, true, b->c, b->stride, (u32)b->n, form, out_kind, d_out};
    a.col0 = (u32)first;
    return msm_dispatch(cx, b->curve, a, st);
}

extern "C" int h2_bases_set_blind_base(h2_bases_t g, const uint64_t *w_xy, int form) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    if (!w_xy || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return set_blind_base_host(*b, w_xy, form);
}

// Two commits from ONE column over a registered basis: column i < n - 4 belongs to output (i >> pair_shift) & 1, the last four
// columns to outputs 0, 1, 0, 1.  The shape of a round of the opening argument written over the original generators
// (poly/commitment/prover.rs:107-114; opening.py): L_j and R_j have disjoint supports in g (the low / high half of every
// 2^(k-j) block), so their scalars share one column, and the basis g || u || u || w || w carries the [value z] U and
// [rand] W terms of each.  One sort, one bucket accumulation into two slices, one fold: ~1.5 ms per round at k = 20 against
// 2.0 ms for two half-empty commits.  d_out receives output 0 then output 1.
// ---- batched commits: the columns of one prover phase (plonk/prover.rs:93-101, 301-313; vanishing/prover.rs:96-108)
// are independent; spread them over internal streams so one column's latency-bound sort / reduce kernels run beside
// another's accumulate, then join on the caller's stream.  Measured at 2^20 (ms per commit):
// 1 stream 1.94; 2 streams 1.64-1.70; 3 streams 1.60-1.64; 4 streams 1.56-1.75.  An accumulate launch fills the register
// file, so the other columns' short kernels run in the gaps between accumulates; narrowing the accumulates
// (lane fraction 0.5) lets them co-reside instead and reaches 1.52 in long runs, but drains badly on short batches.
namespace {
struct BatchStreams {
    std::mutex mu;
    std::vector<hipStream_t> s;
    std::vector<hipEvent_t> done;
    hipEvent_t fork = nullptr;
};
BatchStreams &batch_streams() {
    static BatchStreams b[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return b[dev & 15];
}
}  // namespace

extern "C" int h2_commit_batch_device(h2_bases_t g, const void *const *d_scalars, size_t count, size_t n, const void *d_w_xy,
                                      const void *const *d_blinds, int form, int out_kind, void *const *d_outs, void *stream) {
    if (!d_scalars || !d_outs || (d_w_xy && !d_blinds)) return H2_ERR_ARGS;      // d_blinds without d_w_xy: the handle's blind base
    if (count == 0) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    BatchStreams &bs = batch_streams();
    std::lock_guard<std::mutex> lk(bs.mu);
    const size_t want = std::min<size_t>(3, count);
    const double fraction = 0.0;  // the process-wide option
    while (bs.s.size() < want) {
        hipStream_t st;
        hipEvent_t ev;
        H2_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        H2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        bs.s.push_back(st);
        bs.done.push_back(ev);
    }
    if (!bs.fork) H2_HIP(hipEventCreateWithFlags(&bs.fork, hipEventDisableTiming));
    hipStream_t user = (hipStream_t)stream;
    if (d_blinds) {  // a presented w is checked (and installed if it differs) ONCE, on the caller's stream, before forking
        auto b = find_bases(g);
        if (!b) return H2_ERR_HANDLE;
        if (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) return H2_ERR_ARGS;
        if (d_w_xy) {
            if ((rc = override_blind_base_device(*b, d_w_xy, form, user)) != H2_OK) return rc;
        } else {
            std::lock_guard<std::mutex> bl(b->mu);
            if (!b->blind_set) {
                set_last_error_msg("batch commit with blinds but the handle has no blind base: call h2_bases_set_blind_base, or pass d_w_xy");
                return H2_ERR_ARGS;
            }
        }
    }
    // The column-batched form (ColIn / ColStride above): groups of up to kMaxCols columns, each group ONE launch set with
    // blockIdx.z = column.  A single group runs on the caller's stream as it is; several groups alternate over two internal
    // streams, so that one group's sort and fold run beside the other's accumulate.  Shapes the batched form does not take
    // (narrow windows, small columns: msm_launch says so before launching anything) fall through to one commit per column
    // on three streams, as before.  H2_BATCH_COLS: sweeps only (1 = the per-column form).
    static const int batch_env = [] { const char *e = ab_env("H2_BATCH_COLS"); int v = e ? atoi(e) : 0; return v >= 1 && v <= kMaxCols ? v : 0; }();
    const size_t batch_cols = batch_env ? (size_t)batch_env : (size_t)kMaxCols;
    // Which form (measured on one MI355X, bench/tools/batch_vs_fork.py, profiles/r04_batch_vs_fork.txt; ms per column, batched / forked):
    //   2^13 x 8: 0.059 / 0.116    2^14 x 8: 0.065 / 0.115    2^16 x 8: 0.102 / 0.143    2^18 x 2, 3, 8: 0.353 / 0.386, 0.302 / 0.323, 0.287 / 0.290
    //   2^20 x 2: 1.105 / 1.177    2^20 x 3: 1.094 / 1.099    2^20 x 8: 1.064 / 1.037
    // Below ~2^18 points a commit is a chain of short launches and the batched form shares every one of them; at 2^20 the accumulate
    // is 80 % of a commit, K x 512 workgroups of it do not tile the chip as evenly as one launch per column sized to it, and three
    // streams of whole commits hide more of the tails: many full-size columns keep the forked form.
    const bool prefer_fork = !batch_env && n >= ((size_t)1 << 19) && count > 3;
    if (count >= 2 && batch_cols >= 2 && n > 0 && !prefer_fork) {
        auto b = find_bases(g);
        if (!b) return H2_ERR_HANDLE;
        if (bad_common(b->curve, form, out_kind) || n > b->n) return H2_ERR_ARGS;
        for (size_t i = 0; i < count; ++i)
            if (!d_scalars[i] || !d_outs[i] || (d_blinds && !d_blinds[i])) return H2_ERR_ARGS;
        const size_t groups = (count + batch_cols - 1) / batch_cols;
        const bool forked = groups > 1;
        if (forked) {
            H2_HIP(hipEventRecord(bs.fork, user));
            for (size_t i = 0; i < 2; ++i) H2_HIP(hipStreamWaitEvent(bs.s[i], bs.fork, 0));
        }
        bool taken = true;
        size_t first = 0;
        for (size_t gidx = 0; gidx < groups && rc == H2_OK; ++gidx) {
            const size_t gsize = count / groups + (gidx < count % groups ? 1 : 0);      // balanced groups
            hipStream_t st = forked ? bs.s[gidx & 1] : user;
            MsmContext &cx = msm_ctx(st);
            std::lock_guard<std::mutex> cl(cx.mu);
            MsmArgs a{d_scalars[first], d_blinds ? d_blinds[first] : nullptr, b->d_table, nullptr, n, true, b->c, b->stride, (u32)b->n, form, out_kind, d_outs[first]};
            a.lane_fraction = fraction;
            if (gsize > 1) {
                a.ncols = (int)gsize;
                a.col_scalars = d_scalars + first;
                a.col_blinds = d_blinds ? d_blinds + first : nullptr;
                a.col_outs = d_outs + first;
            }
            rc = msm_dispatch(cx, b->curve, a, st);
            if (rc == H2_ERR_BATCH_SHAPE) {            // nothing was launched
                rc = H2_OK;
                taken = false;
                break;
            }
            first += gsize;
        }
        if (forked)
            for (size_t i = 0; i < 2; ++i) {
                H2_HIP(hipEventRecord(bs.done[i], bs.s[i]));
                H2_HIP(hipStreamWaitEvent(user, bs.done[i], 0));
            }
        if (taken || rc != H2_OK) return rc;
    }
    H2_HIP(hipEventRecord(bs.fork, user));
    for (size_t i = 0; i < want; ++i) H2_HIP(hipStreamWaitEvent(bs.s[i], bs.fork, 0));
    for (size_t i = 0; i < count && rc == H2_OK; ++i)
        rc = commit_device_impl(g, d_scalars[i], n, nullptr, d_blinds ? d_blinds[i] : nullptr, form, out_kind, d_outs[i], bs.s[i % want], true, fraction);
    for (size_t i = 0; i < want; ++i) {
        H2_HIP(hipEventRecord(bs.done[i], bs.s[i]));
        H2_HIP(hipStreamWaitEvent(user, bs.done[i], 0));
    }
    return rc;
}

// Independent multiexps over caller-supplied bases in one call (the L_j / R_j pair of an opening-argument round,
// poly/commitment/prover.rs:107-108): same fork / join as the batched commits, so the latency-bound window combine of
// one overlaps the bucket accumulation of the other.
extern "C" int h2_msm_batch_device(int curve, const void *const *d_scalars, const void *const *d_bases_xy, const size_t *n, size_t count,
                                   int form, int out_kind, void *const *d_outs, void *stream) {
    if (!d_scalars || !d_bases_xy || !n || !d_outs) return H2_ERR_ARGS;
    if (count == 0) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    BatchStreams &bs = batch_streams();
    std::lock_guard<std::mutex> lk(bs.mu);
    const size_t want = std::min<size_t>(3, count);
    while (bs.s.size() < want) {
        hipStream_t st;
        hipEvent_t ev;
        H2_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        H2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        bs.s.push_back(st);
        bs.done.push_back(ev);
    }
    if (!bs.fork) H2_HIP(hipEventCreateWithFlags(&bs.fork, hipEventDisableTiming));
    hipStream_t user = (hipStream_t)stream;
    H2_HIP(hipEventRecord(bs.fork, user));
    for (size_t i = 0; i < want; ++i) H2_HIP(hipStreamWaitEvent(bs.s[i], bs.fork, 0));
    for (size_t i = 0; i < count && rc == H2_OK; ++i)
        rc = h2_msm_device(curve, d_scalars[i], d_bases_xy[i], n[i], form, out_kind, d_outs[i], bs.s[i % want]);
    for (size_t i = 0; i < want; ++i) {
        H2_HIP(hipEventRecord(bs.done[i], bs.s[i]));
        H2_HIP(hipStreamWaitEvent(user, bs.done[i], 0));
    }
    return rc;
}

extern "C" int h2_points_sum_device(int curve, const void *d_points_xyz, size_t count, int form, int out_kind, void *d_out, void *stream) {
    if (bad_common(curve, form, out_kind) || !d_out || (count && !d_points_xyz) || count > (1u << 20)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool mont = form == H2_FORM_MONTGOMERY;
    if (curve == H2_PALLAS) hipLaunchKernelGGL((k_points_sum<FP>), dim3(1), dim3(64), 0, st, (const u32 *)d_points_xyz, (u32)count, (u32 *)d_out, mont, out_kind, mont);
    else hipLaunchKernelGGL((k_points_sum<FQ>), dim3(1), dim3(64), 0, st, (const u32 *)d_points_xyz, (u32)count, (u32 *)d_out, mont, out_kind, mont);
    H2_HIP(hipGetLastError());
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
