// Internal header of the multi-scalar-multiplication translation units (msm_*.hip): constants, the structures that travel between
// the host orchestration and the kernels, small device helpers more than one unit uses, and the declarations of every kernel and
// host function that is defined in one unit and used from another.  Kernels are templates on the base field FB / scalar field FS
// (FP = 0, FQ = 1); the unit that defines one instantiates it explicitly for both curves and the others see `extern template`.
// Nothing here is part of the C ABI (include/halo2_mi355x.h).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "curve_wide.cuh"
#include "curve9.cuh"
#include "curve9_wide.cuh"
#include "glv.cuh"
#include "host_field.h"
namespace h2 {

extern std::atomic<double> g_lane_fraction;      // h2_set_option("msm_lane_fraction")
extern std::atomic<size_t> g_pipe_chunk;         // h2_set_option("host_commit_chunk"): sweeps only
static constexpr int kMaxC = 16;          // generic path (and the 16-bit digit codes of the one-pass sort)
static constexpr int kMaxCShared = 20;    // registered path: one bucket slice, two-pass sort, 32-bit digit codes
static constexpr u32 kZeroCode = 0xFFFFu;
static constexpr u32 kZero32 = 0xFFFFFFFFu;
static constexpr u32 kS1Scalars = 2048;   // scalars per workgroup of the two-pass sort's first pass
static constexpr u32 kS2Chunk = 16384;    // entries per workgroup of its second pass
static constexpr size_t kLdsCap = 160 * 1024 - 512;

// two-pass sort geometries (msm_launch.hip)
bool sort2_geometry(u32 stride, int c, int *lowb_out, int *lb_out, int *side_out = nullptr, u32 *s1_out = nullptr);
bool pair_geometry(size_t m, int c, u32 stride, int *lowb_out, int *lb_out, u32 *nh_out, u32 *s1_out);

static constexpr int kSeg = 4;     // buckets per reduce segment when the fold is latency-bound (few segments), else 2 kSeg
static constexpr u32 kScanBlock = 1024;

struct MsmShape {
    int c, W;
    u32 NB;          // buckets per slice = 2^(c-1)
    u32 slices;      // generic: W; registered (precomputed table): 1
    size_t m;        // digit columns per window = points used + (blind ? 1 : 0)
    size_t items;    // digit codes per slice: generic m, registered W*m
    u32 B, chunk;    // chunks per slice, codes per chunk
    u32 total_buckets;
};

inline bool glv_applies(size_t n) { return n <= ((size_t)1 << 28); }      // the endomorphism split keeps 2n below 2^31
int choose_c(size_t n, bool shared_buckets);
MsmShape make_shape(size_t m, int c, bool shared_buckets, bool glv = false);

__device__ __forceinline__ u32 limb_at(const fe &s, int idx) {
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v = (idx == i) ? s.v[i] : v;
    return v;
}

// The sort and tail stages are short dependent chains; when another stream's msm_accumulate shares the SIMDs they
// must win issue arbitration or they stretch 3-4x and the stream they belong to feeds the chip late (timeline in
// DESIGN.md section 5).  msm_accumulate stays at priority 0.
#ifndef H2_TAIL_PRIO
#define H2_TAIL_PRIO 3      // measured again in round 4 under the bench's load (profiles/r04_ab_tail_priority.txt)
#endif
#define H2_LATENCY_STAGE() __builtin_amdgcn_s_setprio(H2_TAIL_PRIO)

// ---- two-pass sort for the registered (one bucket slice) path ------------------------------------------------------
// A counting sort straight into 2^15 buckets writes 16.8 M 4-byte entries to 16.8 M unrelated places: every workgroup
// keeps 32768 cache lines open, nothing combines in L2, and the pass runs at random-scatter speed (0.2 ms,
// bench/ubench_scatter.hip) on top of 32 MiB of per-chunk histograms.  Split the bucket id instead:
//   pass 1  partition by the top HIB bits (~512 bins): a workgroup keeps one open line per bin, so its 4-byte writes
//           combine in its XCD's L2; the low LOWB bits of the bucket ride in the unused bits of the entry;
//           the digits are recomputed from the scalars (same 32 B per scalar as a digit buffer would cost to read);
//   pass 2  chunks of 16 K entries of the pass-1 output span one or two bins = 64..128 buckets: counting sort inside
//           that window, a few hundred bytes per bucket per chunk.
// Per-chunk histograms shrink from 32 MiB to ~2.4 MiB.  Entry order inside a bucket is irrelevant to the sum.
struct Sort2 {
    u32 m;            // scalars incl. the optional blind
    int c, W, mont;
    u32 stride, extra_col;
    int lowb, lb;     // low bucket bits carried in the entry at bit `lb`
    u32 nh;           // pass-1 bins = NB >> lowb
    u32 B1;           // pass-1 workgroups (kS1Scalars scalars each)
    u32 K2, B2;       // pass-2 chunk size and worst-case chunk count
    u32 lds_window;   // widest pass-2 window (buckets) whose counters fit LDS; wider ones count in HBM
    u32 s1_scalars;   // scalars per pass-1 workgroup (a multiple of 1024)
    u32 run_lanes;    // lanes that copy one bin's run out of the pass-1 stage (a power of two <= 64)
    u32 nb;           // buckets per slice; generic path: sort key = window * nb + bucket, entry = digit column
    int side;         // 1: the low bucket bits of a tagged entry live in the 16-bit side array, not in the entry
    u32 col0;         // registered path: scalar i sits in table column col0 + i (a column RANGE of the table: the chunks of a pipelined host commit)
    int pair_shift;   // >= 0: registered PAIR commit -- column i < pair_n feeds output (i >> pair_shift) & 1, a tail column
    u32 pair_n;       //       i >= pair_n feeds output (i - pair_n) & 1; sort key = side * nb + bucket (two bucket slices)
};

// one GROUP of window slices of a large generic multiexp, sorted on its own (msm_generic.hip; kernels msm_d1_* in msm_sort.hip)
struct GroupSort {
    u32 cols;         // digit columns = 2 x scalars (column i: k1 / P_i, column m + i: k2 / phi(P_i))
    u32 row;          // digit row stride: cols rounded up to 8
    u32 w0, ns;       // the group's window slices [w0, w0 + ns)
    u32 nb;           // buckets per slice; sort key = (w - w0) nb + bucket
    int lowb, lb;     // the low `lowb` key bits ride in the tagged entry at bit `lb` (above the column index)
    u32 nh;           // pass-1 bins = ceil(ns nb >> lowb)
    u32 S;            // digit columns per pass-1 workgroup (a multiple of 8)
    u32 run_lanes;    // lanes that copy one bin's run out of the pass-1 stage
};

// ---- column-batched launches ------------------------------------------------------------------------------------------------------
// The column commits of a prover phase are independent multiexps over ONE registered table (plonk/prover.rs:93-101, 301-313;
// vanishing/prover.rs:96-108).  A batched commit (h2_commit_batch_device) runs every stage ONCE for K columns: blockIdx.z is the
// column, the per-column work areas lie a fixed stride apart, the scalar / blind / output pointers ride in the kernel arguments.
// The sort and fold stages are chains of short latency-bound launches when they serve one column; with K columns per launch they
// become throughput kernels, and the accumulate's K x 512 workgroups refill the chip as they retire instead of as whole launches.
static constexpr int kMaxCols = 8;
struct ColIn {                      // pass 1 of the sort
    const u32 *scalars[kMaxCols];
    const u32 *blinds[kMaxCols];    // null entries: no blind term
};
struct ColOut {                     // fold9_planes
    u32 *out[kMaxCols];
};
struct ColStride {                  // 32-bit words between the areas of consecutive columns (all zero for a single column)
    u32 hist, plan, items, starts, heavy, hscratch, heads, buckets, lines, planes, ctr;
    u32 entries;      // sorted entries: `items` apart, or 0 when the columns are JOINED (below)
    u32 lowprio;      // != 0: the fold kernels keep wave priority 0 (they have slack and must not stretch an accumulate that shares their SIMDs)
    u32 joined;       // != 0: the pass-1 bin count nh.  The columns' sorted entries then form ONE list (column z's behind those of the
                      // columns before it) with ONE boundary array over K x total_buckets buckets (bucket b of column z at
                      // z * total_buckets + b), which msm_accumulate and fold9_finish walk as if it were a single commit
};
#define H2_COLZ(ptr, stride) ((ptr) + (size_t)blockIdx.z * (stride))
// joined columns: the entries of the columns before this one (each column's total sits behind its pass-1 bin starts, at [nh])
__device__ __forceinline__ u32 col_entry_base(const u32 *__restrict__ bin_start_col0, const ColStride &cs) {
    u32 s = 0;
    if (cs.joined)
        for (u32 k = 0; k < blockIdx.z; ++k) s += bin_start_col0[(size_t)k * cs.plan + cs.joined];
    return s;
}

// exclusive scan of v[0 .. n) in LDS by the first wave (n <= 4096); returns the total to every lane of that wave
__device__ __forceinline__ u32 wave0_excl_scan(u32 *v, u32 n) {
    const u32 per = (n + 63) / 64, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    u32 sum = 0;
    for (u32 h = lo; h < hi; ++h) sum += v[h];
    u32 incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(incl, off, 64);
        if ((int)threadIdx.x >= off) incl += t;
    }
    u32 run = incl - sum;
    for (u32 h = lo; h < hi; ++h) {
        u32 t = v[h];
        v[h] = run;
        run += t;
    }
    return __shfl(incl, 63, 64);
}

static constexpr u32 kMaxBig = 32, kBigChunks = 64;      // big bins sorted by the chunked kernels (msm_s2_big_*); workgroups per big bin
static constexpr u32 kS2StageWindow = 3072;

// lanes actually used for M sorted entries: the launch is sized for the worst case (no zero digits); sparse or tiny
// columns use fewer lanes so that a lane's range keeps >= `div` entries
// (16 entries for full-size columns; 8 for small ones, which are latency-bound: more, shorter lanes -- `div`)
__device__ __forceinline__ u32 eff_lanes(u32 M, u32 T, u32 div) { return min(T, max(256u, (M + div - 1) / div)); }

// largest b in [0, n) with arr[b] <= t  (arr non-decreasing, arr[0] = 0)
__device__ __forceinline__ u32 upper_bucket(const u32 *__restrict__ arr, u32 n, u32 t) {
    u32 lo = 0, hi = n;  // invariant: arr[lo] <= t < arr[hi]  (arr[n] = total > t)
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (arr[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// finisher constants: buckets owning more than kHeavy range heads are parked on a list and summed by whole workgroups
static constexpr u32 kHeavy = 64;
static constexpr u32 kMaxHeavy = 512;   // heavy buckets handed to the workgroup path; any beyond that are summed in place
static constexpr u32 kHeavyBlocks = 32;
static constexpr u32 kHeavyRows = 16;     // workgroup rows of the heavy-bucket launches: they walk the list (at most kMaxHeavy long, usually empty)
// ---- the levels after that: trees, on quads of lanes ---------------------------------------------------------------------------
// Past the per-bucket stage every sum is a TREE (a line of the bucket matrix, the heads of a heavy bucket, a bit plane of the
// line sums): its depth in dependent point additions is the latency, and most lanes idle anyway.  A point addition on a quad
// of lanes (curve9_wide.cuh: 4 product levels of ~165 instructions) takes a wave ~850 instructions for 16 additions where
// the one-lane form takes ~2700 for up to 64 -- less wave time from the third tree level up, and a third of the latency at
// every level (a line of 256 buckets: 72 -> ~25 us).
//
static constexpr int kOutSliceSum = 100;      // fold9_planes: the slice's sum as XYZZ (32 words) instead of a finished commitment
#ifndef H2_FOLD_D
#define H2_FOLD_D 2
#endif
// quad q of the workgroup sums the raw points src[36 * index(k)], k = q, q + nq, ... < count.  D points are in flight: each
// lane fetches ONE coordinate (9 words) of each and the quad exchanges them by DPP when the point's turn comes -- the strided
// loads of a line overlap instead of each waiting behind the previous addition.  These kernels run while other streams'
// msm_accumulate holds two waves a SIMD (2 x 168 of 512 registers): they are bounded to the 168 that still fit beside them
// (__launch_bounds__(.., 3)), which two points in flight meet without spilling the addition's own temporaries.
template <int FB, int D = 4, class Index> __device__ __forceinline__ xyzz9<FB> fold9_quad_gather(const u32 *__restrict__ src, u32 count, Index index) {
    const u32 q = threadIdx.x / kGroup, l = threadIdx.x & (kGroup - 1), nq = blockDim.x / kGroup;
    xyzz9<FB> acc = xyzz9_identity<FB>();
    for (u32 k0 = q; k0 < count; k0 += D * nq) {
        fe9 co[D];
#pragma unroll
        for (int j = 0; j < D; j++) {
            const u32 k = k0 + j * nq;
            co[j] = fe9_zero();                                  // (an all-zero point is the identity: skipped by the addition)
            if (k < count) {
                const u32 *w = src + 36 * (size_t)index(k) + 9 * l;
#pragma unroll
                for (int i = 0; i < 9; i++) co[j].v[i] = (i32)w[i];
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++)
            xyzz9_add_wide<FB>(acc, xyzz9<FB>{g9_bcast<0>(co[j]), g9_bcast<1>(co[j]), g9_bcast<2>(co[j]), g9_bcast<3>(co[j])});
    }
    return acc;
}
// the sum of the nq points the quads of a workgroup hold -> quad 0 (every lane of it); sh: nq / 2 raw points.  (Rotating the
// tree by a wave per workgroup index, so that the workgroups sharing a CU do not all finish on their wave 0, was measured: the
// line sums got 5 us SLOWER.)
__device__ __forceinline__ u32 fold9_vquad() { return threadIdx.x / kGroup; }
__device__ __forceinline__ bool fold9_root() { return threadIdx.x < kGroup; }
template <int FB> __device__ __forceinline__ xyzz9<FB> fold9_quads_sum(xyzz9<FB> acc, u32 *sh, u32 first_quads = 0) {      // first_quads (a power of two): only those hold a summand
    const u32 q = fold9_vquad(), nq = first_quads ? first_quads : blockDim.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    for (u32 off = nq / 2; off > 0; off >>= 1) {
        if (q >= off && q < 2 * off && lead) xyzz9_store_raw<FB>(sh + 36 * (size_t)(q - off), acc);
        __syncthreads();
        if (q < off) xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(sh + 36 * (size_t)q));
        __syncthreads();
    }
    return acc;
}
// The Horner step over the window slices of SEVERAL ranges of one multiexp (h2_msm's range pipeline): slice sums are linear in the
// points, so sum_q Horner(S_q) = Horner(sum_q S_q) -- quad w adds slice w's sums over the ranges (side by side), then quad 0 runs ONE
// chain of (slices - 1) c doublings instead of one chain per range.
struct RangeSums {
    const u32 *p[16];
};

// ---- host orchestration: the per-(device, stream) workspace and the argument block of one multiexp ---------------------------------
struct MsmContext {
    std::mutex mu;
    DevBuf digits, hist, counts, starts, bsums, entries, heads, heavy, hscratch, buckets, partial, ssums, stage_s, stage_b,
        out, small, tagged, tagged_low, plan, seg9, bases9, collapse, collapse_list, fold_ctr;
    void release_all() {
        for (DevBuf *b : {&digits, &hist, &counts, &starts, &bsums, &entries, &heads, &heavy, &hscratch, &buckets, &partial, &ssums,
                          &stage_s, &stage_b, &out, &small, &tagged, &tagged_low, &plan, &seg9, &bases9, &collapse, &collapse_list, &fold_ctr})
            b->release();
        if (copy_stream) (void)hipStreamDestroy(copy_stream);      // (h2_trim: the device is idle)
        if (copy_done) (void)hipEventDestroy(copy_done);
        copy_stream = nullptr;
        copy_done = nullptr;
        if (side) (void)hipStreamDestroy(side);
        for (hipEvent_t *e : {&ev_fork, &ev_conv, &ev_acc_a, &ev_join}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
        side = nullptr;
        for (hipStream_t &s : gstream) {
            if (s) (void)hipStreamDestroy(s);
            s = nullptr;
        }
        for (hipEvent_t &e : gev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        gpending = false;
        if (gdone) (void)hipEventDestroy(gdone);
        gdone = nullptr;
    }
    bool attr_set = false, attr2_set = false, attr_bins_set = false, attr_grouped_set = false;
    // the grouped form of a large generic multiexp (msm_generic.hip), latency form: gstream[1] carries the call's accumulates (and whatever is
    // serial with them), gstream[0] and [2] everything that runs beside them (the bases' conversion, the later groups' sorts, the groups' folds and
    // links of the Horner chain) -- three streams created back to back, so that they sit on different hardware queues whatever queue the CALLER's stream shares
    // with whom; gev: fork, conversion done, per group sorted / accumulated / chained, then begin / end (the caller's stream only waits on those)
    static constexpr int kMaxGroups = 4;
    hipStream_t gstream[1 + kMaxGroups] = {};
    hipEvent_t gev[4 + 3 * kMaxGroups] = {};
    // recorded behind every grouped multiexp of this context; other contexts ask it whether a generic multiexp is in flight on ANOTHER stream
    // (msm_other_generic_in_flight): independent calls side by side take the throughput form, a lone call the latency form
    hipEvent_t gdone = nullptr;
    std::atomic<bool> gpending{false};
    hipStream_t copy_stream = nullptr;      // h2_msm: the bases cross PCIe on this one while the sort runs (null-stream context only)
    hipEvent_t copy_done = nullptr;
    // the slice split of a large generic multiexp (msm_launch): the upper slices' fold and Horner chain run on `side` beside the lower
    // slices' accumulate; fork / conv / acc_a / join order the two streams
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_conv = nullptr, ev_acc_a = nullptr, ev_join = nullptr;
    u32 lanes[2][3] = {{0, 0, 0}, {0, 0, 0}};  // resident lanes of msm_accumulate<FP / FQ, plain / GLV> on this device
};

MsmContext &msm_ctx(hipStream_t st = nullptr);          // msm_launch.hip

struct MsmArgs {
    const void *d_scalars;       // n_used scalars
    const void *d_extra_scalar;  // blind or null
    const void *d_bases;         // generic: n_used affine points; registered: table [W][stride]
    const void *d_extra_base;    // generic + blind: w's buffer; else null
    size_t n_used;
    bool table;                  // registered (precomputed) shape
    int c;                       // window bits (fixed by the table when `table`)
    u32 stride;                  // table row stride (n_registered + 1)
    u32 extra_col;               // table column of the blind's base; 0xFFFFFFFF when unused
    int form, out_kind;
    void *d_out;
    double lane_fraction = 0.0;  // 0 = the process-wide option; the batch entry point narrows its commits
    int pair_shift = -1;         // >= 0 (registered only): two outputs from one column, see Sort2::pair_shift; d_out holds both
    u32 pair_n = 0;
    u32 col0 = 0;                // registered only: the scalars are table columns [col0, col0 + n_used)
    // A commit assembled from RANGES (the chunks of a pipelined host transfer): each range runs sort + accumulate + finish and
    // adds its finished buckets into `add_into` (XYZZ, reference Montgomery form, [NB]) instead of folding them; one fold-only
    // call (`fold_from`) then reduces the summed buckets: the ranges share ONE fold instead of paying one each.
    u32 *add_into = nullptr;
    const u32 *fold_from = nullptr;
    // Column-batched commit (registered tables, wide windows): ncols independent columns of n_used scalars each run through ONE
    // launch set, blockIdx.z = column (ColIn / ColOut / ColStride above).  Host arrays of device pointers; col_blinds may be null.
    // msm_launch answers H2_ERR_BATCH_SHAPE before launching anything when the shape does not take the batched form.
    // phase: 0 the whole multiexp; 1 stop after the sort (nothing has read d_bases yet); 2 resume after it (same arguments, same stream).
    // h2_msm uses 1 / 2 to run the sort -- which needs the scalars only -- while the bases are still crossing PCIe.  Its range
    // pipeline cuts finer: 3 = resume after the sort and stop after the accumulate (the full-chip part); 4 = the fold alone, behind a
    // phase-3 call of the same arguments (possibly on ANOTHER stream, ordered by the caller's events).  slice_sums_only (generic
    // path with window slices on the carry-free fold): the fold stops at the per-slice sums in cx.ssums (XYZZ, 32 words per slice) --
    // the caller runs ONE Horner step over the sums of several ranges (msm_combine_ranges) instead of one chain of ~128 doublings
    // per range; H2_ERR_BATCH_SHAPE, before anything is launched, when the shape does not take that form.
    int phase = 0;
    bool slice_sums_only = false;
    int ncols = 1;
    const void *const *col_scalars = nullptr;
    const void *const *col_blinds = nullptr;
    void *const *col_outs = nullptr;
};
static constexpr int H2_ERR_BATCH_SHAPE = -1000;     // internal: never leaves the library


// msm_generic.hip: the grouped form of a large generic multiexp (unregistered bases, endomorphism split, window slices sorted and
// accumulated in groups); H2_ERR_BATCH_SHAPE, before anything is enqueued, when the shape does not take it (msm_launch then runs its own form)
template <int FB, int FS> int msm_generic_grouped(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st);
extern template int msm_generic_grouped<FP, FQ>(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st);
extern template int msm_generic_grouped<FQ, FP>(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st);
bool msm_other_generic_in_flight(const MsmContext *self);                            // msm_launch.hip: a grouped multiexp of another stream of this device has not completed
int msm_dispatch(MsmContext &cx, int curve, const MsmArgs &a, hipStream_t st);      // msm_launch.hip: the whole multiexp (or one phase of it) on `st`
void to_mont_async(int curve, u32 *d, size_t field_elems, hipStream_t st);           // canonical -> Montgomery in place
bool timeline_on();                                                                  // H2_TIMELINE=1 (diagnostic; never changes a result)
void msm_release_host_pipe();                                                        // msm_host.hip (h2_trim)
void msm_release_host_msm_pipe();

// ---- registered bases -------------------------------------------------------------------------------
struct Bases {
    std::mutex mu;
    int curve = 0;
    size_t n = 0;
    int c = 16, W = 16;
    u32 stride = 0;            // n + 1: column n is the blind's base
    void *d_table = nullptr;   // [W][stride] affine Montgomery points, row w = 2^(c*w) * P
    void *d_blind_tmp = nullptr;  // 64-byte staging slot for a blind base that arrives through a host pointer
    int device = 0;            // the HIP device the table lives on (current at registration)
    // `Params::w` (poly/commitment.rs:26-33) belongs to the handle: h2_bases_set_blind_base installs its multiples as column n.
    // blind_set: the column holds SOME w.  blind_host_known: `blind_host` / `blind_form` are the 64 bytes it was installed from
    // (false after a device-pointer override, whose content the host never sees).
    bool blind_set = false, blind_host_known = false;
    int blind_form = 0;
    unsigned char blind_host[64] = {0};
    DevBuf fill_tmp;           // table_fill's staging, kept only by handles that are refilled (bases_refill_device: the opening argument's G' table)
    int glv = 0;               // != 0: an ENDOMORPHISM table (the opening argument's G', served by pair_subdigit_launch only): rows 0 .. glv - 1 are
                               // 2^(16 w) P, rows glv .. 2 glv - 1 their images phi(2^(16 w) P) = (zeta x, y) = [lambda] 2^(16 w) P; W = 2 glv
    // The table is owned here: it goes back to the allocator when the LAST reference drops -- h2_bases_free only removes
    // the handle, so a commit another host thread is still enqueueing (it holds the shared_ptr from find_bases) keeps the
    // memory alive, and every error path of h2_bases_register releases what it had allocated.
    ~Bases() {
        if (!d_table && !d_blind_tmp && !fill_tmp.ptr) return;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        if (d_table) (void)hipFree(d_table);
        if (d_blind_tmp) (void)hipFree(d_blind_tmp);
        fill_tmp.release();
        if (cur != device) (void)hipSetDevice(cur);
    }
};
// registered tables (msm_table.hip)
std::shared_ptr<Bases> find_bases(h2_bases_t h, bool endomorphism = false);
bool bad_common(int curve, int form, int out_kind);
int table_fill(Bases &b, u32 first, u32 count, hipStream_t st, bool keep_tmp = false);
int set_blind_base_host(Bases &b, const void *host_w_xy, int form);
int override_blind_base_device(Bases &b, const void *d_w_xy, int form, hipStream_t st);
bool pair_subdigits_apply(size_t n);                                                 // msm_subdigit.hip

// compile-time A/B knobs of the accumulate (a second build under build/ab/; the defaults are what ships)
#ifndef H2_ACC_LOOP
#define H2_ACC_LOOP 2       // 1: the round-3 loop (gather issued before the point is repacked); 2: point consumed first (round 4) -- A/B builds
#endif
#ifndef H2_ACC9_WAVES
#define H2_ACC9_WAVES 2     // waves per SIMD the M9 accumulate is compiled for: 2, 3 and 4 run the adds equally fast (profiles/r02_ubench_fe9.txt);
                            // at 2 the register file keeps room for the sort / fold kernels of commits on other streams (3 streams: 903 vs 861 M/s)
#endif

// ---- kernels defined in msm_sort.hip -------------------------------------------------------------------------------------------
template <int FS>
__global__ void __launch_bounds__(256) msm_recode(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont);
extern template __global__ void msm_recode<FP>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont);
extern template __global__ void msm_recode<FQ>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont);
template <int FS>
__global__ void __launch_bounds__(256) msm_recode_glv(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont);
extern template __global__ void msm_recode_glv<FP>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont);
extern template __global__ void msm_recode_glv<FQ>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont);
__global__ void __launch_bounds__(1024) msm_count(const uint16_t *__restrict__ digits, u32 *__restrict__ hist,
                                                  size_t items, u32 chunk, u32 NB);
__global__ void __launch_bounds__(256) msm_chunk_prefix(u32 *__restrict__ hist, u32 *__restrict__ counts, u32 NB,
                                                        u32 B, u32 total_buckets);
__global__ void __launch_bounds__(kScanBlock) msm_scan_blocksums(const u32 *__restrict__ counts, u32 *__restrict__ bsums,
                                                                 u32 total);
__global__ void __launch_bounds__(kScanBlock) msm_scan_top(u32 *__restrict__ bsums, u32 nblocks, u32 *__restrict__ grand);
__global__ void __launch_bounds__(kScanBlock) msm_scan_apply(const u32 *__restrict__ counts, const u32 *__restrict__ bsums,
                                                             const u32 *__restrict__ grand, u32 *__restrict__ starts,
                                                             u32 total);
__global__ void __launch_bounds__(1024) msm_scatter(const uint16_t *__restrict__ digits, const u32 *__restrict__ hist,
                                                    const u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    size_t items, u32 chunk, u32 NB, u32 m, u32 stride, u32 extra_col,
                                                    int table, u32 col0);
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_count(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
extern template __global__ void msm_s1_count<FP, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
extern template __global__ void msm_s1_count<FP, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
extern template __global__ void msm_s1_count<FQ, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
extern template __global__ void msm_s1_count<FQ, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_scatter(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
extern template __global__ void msm_s1_scatter<FP, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
extern template __global__ void msm_s1_scatter<FP, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
extern template __global__ void msm_s1_scatter<FQ, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
extern template __global__ void msm_s1_scatter<FQ, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
__global__ void __launch_bounds__(1024) msm_s1_prefix(u32 *__restrict__ hist1, u32 *__restrict__ bin_count, u32 B1, u32 nh, u32 *__restrict__ z2,
                                                      u32 *__restrict__ z1, u32 *__restrict__ sentinel, ColStride cs);
__global__ void __launch_bounds__(kScanBlock) msm_s2_plan(const u32 *__restrict__ bin_start, Sort2 P, u32 *__restrict__ hlo,
                                                          u32 *__restrict__ woff);
__global__ void __launch_bounds__(1024) msm_s2_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                     const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ hist2);
__global__ void __launch_bounds__(1024) msm_s2_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                       const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P,
                                                       u32 *__restrict__ hist2, const u32 *__restrict__ starts, u32 *__restrict__ entries);
__global__ void __launch_bounds__(1024) msm_s2_bins(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                    Sort2 P, u32 total_buckets, u32 cap, u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    u32 *__restrict__ big, u32 max_big, u32 *__restrict__ zero9, ColStride cs);
__global__ void __launch_bounds__(1024) msm_s2_big_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                         Sort2 P, const u32 *__restrict__ big, u32 *__restrict__ gcnt, ColStride cs);
__global__ void __launch_bounds__(1024) msm_s2_big_prefix(const u32 *__restrict__ bin_start, Sort2 P, u32 total_buckets, const u32 *__restrict__ big,
                                                          u32 *__restrict__ gcnt, u32 *__restrict__ starts, ColStride cs);
__global__ void __launch_bounds__(1024) msm_s2_big_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                           Sort2 P, const u32 *__restrict__ big, const u32 *__restrict__ gcnt, u32 *__restrict__ entries,
                                                           ColStride cs);
__global__ void __launch_bounds__(256) msm_s2_prefix(u32 *__restrict__ hist2, const u32 *__restrict__ bin_start, const u32 *__restrict__ hlo,
                                                     const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ counts, u32 NB);
template <int FS>
__global__ void __launch_bounds__(256) msm_glv_digits(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W, int mont);
extern template __global__ void msm_glv_digits<FP>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W, int mont);
extern template __global__ void msm_glv_digits<FQ>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W, int mont);
__global__ void __launch_bounds__(512) msm_d1_count(const uint16_t *__restrict__ digits, GroupSort P, u32 *__restrict__ hist1);
__global__ void __launch_bounds__(512) msm_d1_scatter(const uint16_t *__restrict__ digits, GroupSort P, const u32 *__restrict__ hist1,
                                                      const u32 *__restrict__ bin_count, u32 *__restrict__ bin_start, u32 *__restrict__ tagged);

// ---- kernels defined in msm_accumulate.hip -------------------------------------------------------------------------------------
template <int FB>
__global__ void __launch_bounds__(256) msm_bases_to_m9_glv(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n);
extern template __global__ void msm_bases_to_m9_glv<FP>(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n);
extern template __global__ void msm_bases_to_m9_glv<FQ>(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n);
// BLOCK: lanes per workgroup.  256 everywhere (two workgroups per CU) except the grouped generic multiexp (msm_generic.hip), whose accumulates
// are launched while the previous group's fold kernels hold wave slots on some CUs: with 256-lane workgroups the dispatcher then puts THREE
// accumulate workgroups on the free CUs (168 registers allow three waves per SIMD), those run at two thirds of the speed for the whole launch
// and the launch ends when they do (331 -> 477 us measured: profiles/r06_generic_grouped.txt).  One 512-lane workgroup per CU cannot double up.
template <int FB, bool GLV, bool M9 = false, int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK, (M9 ? H2_ACC9_WAVES * 256 / BLOCK : 4)) msm_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FP, false, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FP, true, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FP, false, true>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FQ, false, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FQ, true, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FQ, false, true>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FP, false, true, 512>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void msm_accumulate<FQ, false, true, 512>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template <int FB>
__global__ void __launch_bounds__(256) msm_segments_to_r256(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets);
extern template __global__ void msm_segments_to_r256<FP>(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets);
extern template __global__ void msm_segments_to_r256<FQ>(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets);

// ---- kernels defined in msm_fold.hip -------------------------------------------------------------------------------------------
template <int FB>
__global__ void __launch_bounds__(256) msm_finish_buckets(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div);
extern template __global__ void msm_finish_buckets<FP>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div);
extern template __global__ void msm_finish_buckets<FQ>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div);
template <int FB>
__global__ void __launch_bounds__(256) msm_finish_heavy(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div);
extern template __global__ void msm_finish_heavy<FP>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div);
extern template __global__ void msm_finish_heavy<FQ>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div);
template <int FB>
__global__ void __launch_bounds__(64) msm_finish_heavy2(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy);
extern template __global__ void msm_finish_heavy2<FP>(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy);
extern template __global__ void msm_finish_heavy2<FQ>(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy);
template <int FB>
__global__ void __launch_bounds__(256) msm_bucket_add(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb);
extern template __global__ void msm_bucket_add<FP>(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb);
extern template __global__ void msm_bucket_add<FQ>(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb);
template <int FB>
__global__ void __launch_bounds__(256) fold9_finish(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void fold9_finish<FP>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void fold9_finish<FQ>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_finish_heavy(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void fold9_finish_heavy<FP>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
extern template __global__ void fold9_finish_heavy<FQ>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template <int FB>
__global__ void __launch_bounds__(64, 3) fold9_finish_heavy2(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs);
extern template __global__ void fold9_finish_heavy2<FP>(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs);
extern template __global__ void fold9_finish_heavy2<FQ>(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs);
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_rowcol(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs);
extern template __global__ void fold9_rowcol<FP>(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs);
extern template __global__ void fold9_rowcol<FQ>(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs);
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_planes(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs);
extern template __global__ void fold9_planes<FP>(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs);
extern template __global__ void fold9_planes<FQ>(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs);
template <int FB>
__global__ void __launch_bounds__(256) msm_reduce_segments(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg);
extern template __global__ void msm_reduce_segments<FP>(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg);
extern template __global__ void msm_reduce_segments<FQ>(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg);
template <int FB>
__global__ void __launch_bounds__(256) msm_sum_slice(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share);
extern template __global__ void msm_sum_slice<FP>(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share);
extern template __global__ void msm_sum_slice<FQ>(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share);
template <int FB>
__global__ void __launch_bounds__(256) msm_rowcol_sums(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR);
extern template __global__ void msm_rowcol_sums<FP>(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR);
extern template __global__ void msm_rowcol_sums<FQ>(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR);
template <int FB>
__global__ void __launch_bounds__(64) msm_combine(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl = 0, const u32 *__restrict__ addend = nullptr, int addend_first = 0);
extern template __global__ void msm_combine<FP>(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl, const u32 *__restrict__ addend, int addend_first);
extern template __global__ void msm_combine<FQ>(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl, const u32 *__restrict__ addend, int addend_first);
template <int FB>
__global__ void __launch_bounds__(64) msm_combine_ranges(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont);
extern template __global__ void msm_combine_ranges<FP>(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont);
extern template __global__ void msm_combine_ranges<FQ>(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont);

}  // namespace h2
