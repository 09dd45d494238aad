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

static std::atomic<double> g_lane_fraction{1.0};
static std::atomic<size_t> g_pipe_chunk{0};   // h2_set_option("host_commit_chunk"): sweeps only
static constexpr int kMaxC = 16;          // generic path (and the 16-bit digit codes of the one-pass sort)
static constexpr int kMaxCShared = 20;    // registered path: one bucket slice, two-pass sort, 32-bit digit codes
static constexpr u32 kZeroCode = 0xFFFFu;
static constexpr u32 kZero32 = 0xFFFFFFFFu;
static constexpr u32 kS1Scalars = 2048;   // scalars per workgroup of the two-pass sort's first pass
static constexpr u32 kS2Chunk = 16384;    // entries per workgroup of its second pass
static constexpr size_t kLdsCap = 160 * 1024 - 512;

// Two-pass sort geometry for a table of `stride` columns and window width c: the low `lowb` bucket bits ride in the
// entry above the table index (`lb` bits); the remaining bucket_bits - lowb bits select the pass-1 bin.
static bool sort2_geometry(u32 stride, int c, int *lowb_out, int *lb_out, int *side_out = nullptr, u32 *s1_out = nullptr) {
    const int W = 255 / c + 1, bucket_bits = c - 1;
    const uint64_t top = (uint64_t)W * stride - 1;
    if (top >= ((uint64_t)1 << 31)) return false;
    int lb = 0;
    while ((top >> lb) != 0) ++lb;
    // 512 pass-1 bins whenever the low bucket bits have somewhere to ride: in the entry's spare bits above the table index, or
    // -- windows of 18 bits and more at 2^20 points, where those are too few -- in a 16-bit SIDE array next to the tagged list
    // (pass 1 writes 6 bytes per entry instead of 4; without it c = 20 meant 4096 bins and 6-entry runs)
    int lowb = std::min(31 - lb, bucket_bits - 9), side = 0;
    static const bool side_ok = [] { const char *e = getenv("H2_SORT_SIDE"); return !(e && atoi(e) == 0); }();      // A/B switch
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
    static const u32 s1_env = [] { const char *e = getenv("H2_S1_SCALARS"); int v = e ? atoi(e) : 0; return (u32)(v == 512 || v == 1024 || v == 2048 ? v : 0); }();   // sweeps only
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
static bool pair_geometry(size_t m, int c, u32 stride, int *lowb_out, int *lb_out, u32 *nh_out, u32 *s1_out) {
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

// window width: minimise mixed adds + reduce work.  `shared_buckets`: registered bases (one slice).
// The generic path always splits scalars with the endomorphism (glv.cuh): the window Horner it halves (0.35 ms) and the
// smaller fold (9 slices instead of 16) outweigh the extra bucket additions (2n x 9 windows against n x 16, and the
// on-the-fly phi) at every size measured, 2.76 against 3.09 ms even at 2^20.  The size cap only keeps 2n below 2^31.
static bool glv_applies(size_t n) { return n <= ((size_t)1 << 28); }

static int choose_c(size_t n, bool shared_buckets) {
    auto feasible = [&](int c) {
        if (c <= kMaxC) return true;
        int lowb, lb;
        return shared_buckets && n + 1 < ((size_t)1 << 31) && sort2_geometry((u32)n + 1, c, &lowb, &lb);
    };
    if (const char *e = getenv("H2_MSM_C")) {   // tuning sweeps only; the only way to windows beyond 16 bits (see below)
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
static MsmShape make_shape(size_t m, int c, bool shared_buckets, bool glv = false) {
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

// ---- recode: scalars -> signed window digits -------------------------------------------------
// code = 0xFFFF for digit 0, else (|d| - 1) | (d < 0 ? 0x8000 : 0);  digits[w * m + i]
template <int FS>
__global__ void __launch_bounds__(256) msm_recode(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont) {
    H2_LATENCY_STAGE();
    // m counts the optional extra (blind) scalar, which is column m - 1 and lives in its own buffer
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fe s = (extra_scalar && i == m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
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
            code = ((1u << c) - raw - 1) | 0x8000u;   // d = raw - 2^c <= 0: |d| - 1 (d = 0 wraps to 0xFFFF)
        } else {
            carry = 0;
            code = raw ? raw - 1 : kZeroCode;
        }
        digits[(size_t)w * m + i] = (uint16_t)code;
    }
}

// ---- recode with the endomorphism split (generic path): scalar i yields two columns of signed digits, column i for k1
// (base P_i) and column m + i for k2 (base phi(P_i)); half as many windows, so the final Horner over windows needs 128
// doublings instead of 255.  digits[w * 2m + col]
template <int FS>
__global__ void __launch_bounds__(256) msm_recode_glv(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont) {
    H2_LATENCY_STAGE();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fe s = fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_from_mont<FS>(s);
    u32 mag[2][5], neg[2];
    glv_split<FS>(s, mag[0], neg[0], mag[1], neg[1]);
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
    const size_t M = 2 * (size_t)m;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 carry = 0;
        for (int w = 0; w < W; ++w) {
            const int bit = w * c, word = bit >> 5, sh = bit & 31;
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                lo = (word == q) ? mag[part][q] : lo;
                hi = (word + 1 == q) ? mag[part][q] : hi;
            }
            const u64 two = (u64)lo | ((u64)hi << 32);
            const u32 raw = ((u32)(two >> sh) & mask) + carry;
            // digits of |k_part| in [-half + 1, half] (or [-half, half - 1] when the part is negative, so that after the
            // sign flip every digit is again in the encodable set: the code has no room for -half)
            const bool up = neg[part] ? raw >= half : raw > half;
            carry = up;
            const u32 digit_mag = up ? (1u << c) - raw : raw;     // 0 when raw = 0 or raw = 2^c
            const u32 negative = (up ? 1u : 0u) ^ neg[part];
            const u32 code = digit_mag ? ((digit_mag - 1) | (negative ? 0x8000u : 0u)) : kZeroCode;
            digits[(size_t)w * M + (size_t)part * m + i] = (uint16_t)code;
        }
    }
}

// ---- count: LDS histogram per (slice, chunk) ---------------------------------------------------
__global__ void __launch_bounds__(1024) msm_count(const uint16_t *__restrict__ digits, u32 *__restrict__ hist,
                                                  size_t items, u32 chunk, u32 NB) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 h[];
    const u32 b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) h[j] = 0;
    __syncthreads();
    size_t lo = (size_t)b * chunk, hi = min(items, lo + chunk);
    const uint16_t *d = digits + (size_t)sl * items;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) atomicAdd(&h[code & 0x7FFFu], 1u);
    }
    __syncthreads();
    u32 *dst = hist + ((size_t)sl * B + b) * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) dst[j] = h[j];
}

// ---- scan a: per-bucket totals, chunk slices become exclusive prefixes -------------------------
__global__ void __launch_bounds__(256) msm_chunk_prefix(u32 *__restrict__ hist, u32 *__restrict__ counts, u32 NB,
                                                        u32 B, u32 total_buckets) {
    H2_LATENCY_STAGE();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_buckets) return;
    u32 sl = g / NB, j = g % NB;
    u32 run = 0;
    for (u32 b = 0; b < B; ++b) {
        size_t k = ((size_t)sl * B + b) * NB + j;
        u32 t = hist[k];
        hist[k] = run;
        run += t;
    }
    counts[g] = run;
}

// ---- scan b: exclusive scan of the bucket totals -> entry offsets (starts[total] = M) ----------------
// three kernels: block sums, scan of block sums, apply.
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *sh, u32 &total) {
    // blockDim.x == kScanBlock
    const u32 t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (u32 off = 1; off < kScanBlock; off <<= 1) {
        u32 x = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    total = sh[kScanBlock - 1];
    u32 r = sh[t] - v;
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_blocksums(const u32 *__restrict__ counts, u32 *__restrict__ bsums,
                                                                 u32 total) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 g = blockIdx.x * kScanBlock + threadIdx.x;
    u32 tot;
    (void)block_excl_scan(g < total ? counts[g] : 0, sh, tot);
    if (threadIdx.x == 0) bsums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_top(u32 *__restrict__ bsums, u32 nblocks, u32 *__restrict__ grand) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 carry = 0;
    for (u32 base = 0; base < nblocks; base += kScanBlock) {
        u32 i = base + threadIdx.x;
        u32 v = i < nblocks ? bsums[i] : 0, tot;
        u32 ex = block_excl_scan(v, sh, tot);
        if (i < nblocks) bsums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand = carry;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_apply(const u32 *__restrict__ counts, const u32 *__restrict__ bsums,
                                                             const u32 *__restrict__ grand, u32 *__restrict__ starts,
                                                             u32 total) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 g = blockIdx.x * kScanBlock + threadIdx.x;
    u32 tot;
    u32 a = block_excl_scan(g < total ? counts[g] : 0, sh, tot) + bsums[blockIdx.x];
    if (g < total) starts[g] = a;
    if (g == 0) {
        starts[total] = *grand;
        starts[total + 1] = 0xFFFFFFFFu;      // sentinel: msm_accumulate reads the boundary after next without a bounds check
    }
}

// ---- scatter: bucket-sorted entry list ----------------------------------------------------------
// entry = base index | sign << 31.  generic: base index = column (column m-1 of a blinded commit maps to
// `extra_col`); registered: base index = w * stride + column, straight into the precomputed table.
__global__ void __launch_bounds__(1024) msm_scatter(const uint16_t *__restrict__ digits, const u32 *__restrict__ hist,
                                                    const u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    size_t items, u32 chunk, u32 NB, u32 m, u32 stride, u32 extra_col,
                                                    int table, u32 col0) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 off[];
    const u32 b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
    const u32 *src = hist + ((size_t)sl * B + b) * NB;
    const u32 *st = starts + (size_t)sl * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) off[j] = st[j] + src[j];
    __syncthreads();
    size_t lo = (size_t)b * chunk, hi = min(items, lo + chunk);
    const uint16_t *d = digits + (size_t)sl * items;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) {
            u32 pos = atomicAdd(&off[code & 0x7FFFu], 1u);
            const u32 i32 = (u32)i;  // items < 2^31
            u32 w = table ? i32 / m : 0, col = table ? i32 % m : i32;
            if (col == m - 1 && extra_col != 0xFFFFFFFFu) col = extra_col;
            else col += col0;
            entries[pos] = (w * stride + col) | ((code & 0x8000u) << 16);
        }
    }
}

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

// signed window digits of one scalar (same recoding as msm_recode, 32-bit codes so that windows may exceed 16 bits),
// handed to f(w, code): code = kZero32 for digit 0, else (|d| - 1) | (d < 0) << 31
template <int C, typename Fn> __device__ __forceinline__ void for_each_digit_static(const fe &s, Fn f) {
    constexpr int W = 255 / C + 1;
    constexpr u32 mask = (1u << C) - 1, half = 1u << (C - 1), full = 1u << C;
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        constexpr int dummy = 0;
        (void)dummy;
        const int bit = w * C, word = bit >> 5, sh = bit & 31;     // compile-time after unrolling: plain register picks
        u32 v = s.v[word] >> sh;
        if (sh + C > 32 && word + 1 < 8) v |= s.v[word + 1] << (32 - sh);
        const u32 raw = (v & mask) + carry;
        carry = raw > half;
        const u32 code = carry ? (raw == full ? kZero32 : ((full - raw - 1) | 0x80000000u)) : (raw ? raw - 1 : kZero32);   // raw = 2^C: digit 0, carry out
        f(w, code);
    }
}
template <typename Fn> __device__ __forceinline__ void for_each_digit(const fe &s, int c, int W, Fn f) {
    // compile-time widths: the limb picks become plain register selects (the generic loop below indexes the limbs dynamically: at
    // c = 17 the pass-1 kernels took 29 + 78 us against 16 + 53 us for the unrolled c = 20)
    if (c == 16) { for_each_digit_static<16>(s, f); return; }
    if (c == 17) { for_each_digit_static<17>(s, f); return; }
    if (c == 18) { for_each_digit_static<18>(s, f); return; }
    if (c == 19) { for_each_digit_static<19>(s, f); return; }
    if (c == 20) { for_each_digit_static<20>(s, f); return; }
    if (c == 13) { for_each_digit_static<13>(s, f); return; }
    u32 carry = 0;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1), full = 1u << c;
    for (int w = 0; w < W; ++w) {
        int bit = w * c, word = bit >> 5, sh = bit & 31;
        u64 two = (u64)limb_at(s, word) | ((u64)limb_at(s, word + 1) << 32);
        const u32 raw = ((u32)(two >> sh) & mask) + carry;
        carry = raw > half;
        const u32 code = carry ? (raw == full ? kZero32 : ((full - raw - 1) | 0x80000000u)) : (raw ? raw - 1 : kZero32);
        f(w, code);
    }
}

// every non-zero window digit of scalar i as f(sort key, entry base index, sign << 31).
// Registered path: key = bucket (one slice), base = w * stride + column.  Generic path (GLV): the scalar is split
// (glv.cuh), key = w * nb + bucket (slice-major), base = digit column (i for k1 / P_i, m + i for k2 / phi(P_i)).
template <int FS, bool GLV, typename Fn>
__device__ __forceinline__ void emit_entries(const fe &s, u32 i, const Sort2 &P, Fn f) {
    if (!GLV) {
        u32 col = P.col0 + i;
        if (i == P.m - 1 && P.extra_col != 0xFFFFFFFFu) col = P.extra_col;
        u32 side_key = 0;
        if (P.pair_shift >= 0) side_key = (i < P.pair_n ? (i >> P.pair_shift) & 1u : (i - P.pair_n) & 1u) * P.nb;
        for_each_digit(s, P.c, P.W, [&](int w, u32 code) {
            if (code != kZero32) f(side_key + (code & 0x7FFFFFFFu), (u32)w * P.stride + col, code & 0x80000000u, (u32)w);
        });
        return;
    }
    u32 mag[2][5], neg[2];
    glv_split<FS>(s, mag[0], neg[0], mag[1], neg[1]);
    const int c = P.c;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 carry = 0;
        for (int w = 0; w < P.W; ++w) {
            const int bit = w * c, word = bit >> 5, sh = bit & 31;
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                lo = (word == q) ? mag[part][q] : lo;
                hi = (word + 1 == q) ? mag[part][q] : hi;
            }
            const u32 raw = ((u32)(((u64)lo | ((u64)hi << 32)) >> sh) & mask) + carry;
            const bool up = neg[part] ? raw >= half : raw > half;      // same digit set as msm_recode_glv
            carry = up;
            const u32 digit_mag = up ? (1u << c) - raw : raw;
            if (digit_mag) f((u32)w * P.nb + digit_mag - 1, (u32)part * P.m + i, (((up ? 1u : 0u) ^ neg[part]) << 31), (u32)w);
        }
    }
}

// pass 1, COUNT: hist1[blk][h] = this workgroup's entries per bin
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_count(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [nh] counters
    if (gridDim.z > 1) {
        scalars = ci.scalars[blockIdx.z];
        extra_scalar = ci.blinds[blockIdx.z];
        hist1 = H2_COLZ(hist1, cs.hist);
    }
    const u32 nh = P.nh, blk = blockIdx.x;
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) sh[h] = 0;
    __syncthreads();
    for (u32 loc = threadIdx.x; loc < P.s1_scalars; loc += blockDim.x) {
        const u32 i = blk * P.s1_scalars + loc;
        if (i >= P.m) break;
        fe s = (extra_scalar && i == P.m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
        if (P.mont) s = fe_redc<FS>(s);
        emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32, u32, u32) { atomicAdd(&sh[key >> P.lowb], 1u); });
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) hist1[(size_t)blk * nh + h] = sh[h];
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

// pass 1, SCATTER: the workgroup's entries are first grouped by bin in LDS (its per-bin counts are known from the
// count pass), then every bin's run goes out as one contiguous copy -- scattered 4-byte stores issue one lane per
// clock and were the cost of this pass.  hist1 holds the exclusive prefix over workgroups by now.
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_scatter(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    if (gridDim.z > 1) {
        scalars = ci.scalars[blockIdx.z];
        extra_scalar = ci.blinds[blockIdx.z];
        hist1 = H2_COLZ(hist1, cs.hist);
        bin_count = H2_COLZ(bin_count, cs.plan);
        bin_start = H2_COLZ(bin_start, cs.plan);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
    }
    const u32 nh = P.nh, blk = blockIdx.x, B1 = gridDim.x;
    u32 *gstart = sh;                 // [nh] bin_start, then bin_start + this workgroup's offset inside the bin
    u32 *lstart = sh + nh;            // [nh + 1] where the bin's run begins in the stage
    u32 *cursor = lstart + nh + 1;    // [nh]
    u32 *stage = cursor + nh;         // [s1_scalars * digits per scalar] entries
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        const u32 mine = hist1[(size_t)blk * nh + h];
        const u32 next = blk + 1 < B1 ? hist1[(size_t)(blk + 1) * nh + h] : bin_count[h];
        gstart[h] = bin_count[h];
        lstart[h] = next - mine;      // this workgroup's entries in bin h
        cursor[h] = mine;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 M = wave0_excl_scan(gstart, nh);
        const u32 L = wave0_excl_scan(lstart, nh);
        if (threadIdx.x == 0) {
            lstart[nh] = L;
            if (blk == 0) bin_start[nh] = M;
        }
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        if (blk == 0) bin_start[h] = gstart[h];
        gstart[h] += cursor[h];
        cursor[h] = lstart[h];
    }
    __syncthreads();
    const u32 lowmask = (1u << P.lowb) - 1;
    for (u32 loc = threadIdx.x; loc < P.s1_scalars; loc += blockDim.x) {
        const u32 i = blk * P.s1_scalars + loc;
        if (i >= P.m) break;
        fe s = (extra_scalar && i == P.m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
        if (P.mont) s = fe_redc<FS>(s);
        if (!GLV && P.side) {
            // the stage word keeps what the copy-out needs to rebuild the entry: scalar (11 bits, s1_scalars <= 2048), window
            // (6 bits) and the low bucket bits (<= 14), so entry and low bits leave as two contiguous runs per bin
            emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32, u32 sign, u32 w) {
                const u32 pos = atomicAdd(&cursor[key >> P.lowb], 1u);
                stage[pos] = loc | (w << 11) | ((key & lowmask) << 17) | sign;
            });
        } else {
            emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32 base, u32 sign, u32) {
                const u32 pos = atomicAdd(&cursor[key >> P.lowb], 1u);
                stage[pos] = base | ((key & lowmask) << P.lb) | sign;
            });
        }
    }
    __syncthreads();
    // P.run_lanes (16) lanes per bin at a time: contiguous LDS run -> contiguous global run.  A workgroup's run of a bin is ~30
    // entries at 2048 scalars x 15 digits over 1024 bins and a bin costs its group three dependent LDS reads before the first
    // store, whatever the run's length: the loop is as long as the bins a group walks (whole waves: 2x; half waves -> 16 lanes: -10 us)
    const u32 kRunLanes = P.run_lanes, wave = threadIdx.x / kRunLanes, lane = threadIdx.x & (kRunLanes - 1), nwaves = blockDim.x / kRunLanes;
    if (!GLV && P.side) {
        for (u32 h = wave; h < nh; h += nwaves) {
            const u32 l0 = lstart[h], l1 = lstart[h + 1];
            u32 *dst = tagged + gstart[h];
            uint16_t *dlo = tagged_low + gstart[h];
            for (u32 q = l0 + lane; q < l1; q += kRunLanes) {
                const u32 word = stage[q], i = blk * P.s1_scalars + (word & 2047u), w = (word >> 11) & 63u;
                u32 col = P.col0 + i;
                if (i == P.m - 1 && P.extra_col != 0xFFFFFFFFu) col = P.extra_col;
                dst[q - l0] = (w * P.stride + col) | (word & 0x80000000u);
                dlo[q - l0] = (uint16_t)((word >> 17) & 0x3FFFu);
            }
        }
        return;
    }
    for (u32 h = wave; h < nh; h += nwaves) {
        const u32 l0 = lstart[h], l1 = lstart[h + 1];
        u32 *dst = tagged + gstart[h];
        for (u32 q = l0 + lane; q < l1; q += kRunLanes) dst[q - l0] = stage[q];
    }
}

// column-wise exclusive scan of hist1[B1][nh]; bin_count[h] = column total.  16 columns per workgroup, 64 row groups.
// (It also zeroes the two small counter areas later kernels of the same multiexp count into -- `z2`: the two words of the
// heavy-bucket list, `z1`: the oversized-bin counter of pass 2 -- which saves two 5 us memset nodes on the stream.)
__global__ void __launch_bounds__(1024) msm_s1_prefix(u32 *__restrict__ hist1, u32 *__restrict__ bin_count, u32 B1, u32 nh, u32 *__restrict__ z2,
                                                      u32 *__restrict__ z1, u32 *__restrict__ sentinel, ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ u32 part[64][17];
    if (gridDim.z > 1) {
        hist1 = H2_COLZ(hist1, cs.hist);
        bin_count = H2_COLZ(bin_count, cs.plan);
        z2 = H2_COLZ(z2, cs.heavy);
        if (z1) z1 = H2_COLZ(z1, cs.plan);
        sentinel = H2_COLZ(sentinel, cs.starts);
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        if (threadIdx.x < 2) z2[threadIdx.x] = 0;
        else if (threadIdx.x == 2) { if (z1) z1[0] = 0; }
        else *sentinel = 0xFFFFFFFFu;       // starts[total_buckets + 1]: what msm_accumulate reads past the last boundary -- written HERE for
                                            // every form of pass 2 (one-launch bins, oversized bins, chunked); the one-pass sort: msm_scan_apply
    }
    const u32 r = threadIdx.x >> 4, cl = threadIdx.x & 15, col = blockIdx.x * 16 + cl;
    const u32 rg = (B1 + 63) / 64, r0 = min(B1, r * rg), r1 = min(B1, r0 + rg);
    u32 sum = 0;
    if (col < nh)
        for (u32 row = r0; row < r1; ++row) sum += hist1[(size_t)row * nh + col];
    part[r][cl] = sum;
    __syncthreads();
    u32 run = 0;
    for (u32 q = 0; q < r; ++q) run += part[q][cl];
    if (col < nh) {
        for (u32 row = r0; row < r1; ++row) {
            const size_t k = (size_t)row * nh + col;
            const u32 t = hist1[k];
            hist1[k] = run;
            run += t;
        }
        if (r == 63) bin_count[col] = run;
    }
}

// largest h in [0, nh) with bin_start[h] <= p   (bin_start non-decreasing, bin_start[0] = 0 <= p < bin_start[nh])
__device__ __forceinline__ u32 bin_of(const u32 *__restrict__ bin_start, u32 nh, u32 p) {
    u32 lo = 0, hi = nh;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (bin_start[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

// pass-2 plan: first bin and histogram offset of every chunk (window = the bins the chunk touches); one workgroup
__global__ void __launch_bounds__(kScanBlock) msm_s2_plan(const u32 *__restrict__ bin_start, Sort2 P, u32 *__restrict__ hlo,
                                                          u32 *__restrict__ woff) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    const u32 M = bin_start[P.nh];
    u32 carry = 0;
    for (u32 base = 0; base < P.B2; base += kScanBlock) {
        const u32 cidx = base + threadIdx.x;
        u32 size = 0, h0 = 0;
        if (cidx < P.B2 && (size_t)cidx * P.K2 < M) {
            const u32 p0 = cidx * P.K2, p1 = min(M, p0 + P.K2) - 1;
            h0 = bin_of(bin_start, P.nh, p0);
            size = (bin_of(bin_start, P.nh, p1) - h0 + 1) << P.lowb;
        }
        u32 tot;
        const u32 ex = block_excl_scan(size, sh, tot);
        if (cidx < P.B2) {
            hlo[cidx] = h0;
            woff[cidx] = carry + ex;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) woff[P.B2] = carry;
}

// bin (relative to the chunk's first bin) of list position p; bounds[q] = bin_start[h0 + q + 1]
__device__ __forceinline__ u32 rel_bin(const u32 *bounds, u32 nbins, u32 p) {
    if (p < bounds[0]) return 0;
    u32 lo = 0, hi = nbins - 1;                                 // invariant: bounds[lo] <= p < bounds[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (bounds[mid] <= p) lo = mid;
        else hi = mid;
    }
    return hi;
}

// low bucket bits of tagged entry p: in the entry's spare bits, or in the side array
__device__ __forceinline__ u32 s2_low(const Sort2 &P, const uint16_t *__restrict__ low, u32 p, u32 e, u32 lowmask) {
    return P.side ? (u32)low[p] : (e >> P.lb) & lowmask;
}

// pass 2, COUNT over one chunk of the tagged list: hist2[woff[c] + (bucket - window base)]
__global__ void __launch_bounds__(1024) msm_s2_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                     const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ hist2) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [window] counters, then the window's bin boundaries
    const u32 cidx = blockIdx.x, M = bin_start[P.nh];
    const size_t p0 = (size_t)cidx * P.K2;
    if (p0 >= M) return;
    const u32 p1 = (u32)min((size_t)M, p0 + P.K2);
    const u32 h0 = hlo[cidx], wo = woff[cidx], wsize = woff[cidx + 1] - wo, nbins = wsize >> P.lowb;
    const u32 lowmask = (1u << P.lowb) - 1;
    if (wsize > P.lds_window) {
        // a very sparse column: the chunk's window does not fit LDS.  Few entries by construction -- count in HBM
        // (hist2 is zeroed before this kernel).
        const u32 *gb = bin_start + h0 + 1;
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            atomicAdd(&hist2[wo + ((rel_bin(gb, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask))], 1u);
        }
        return;
    }
    u32 *bounds = sh + wsize;
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) sh[k] = 0u;
    for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];
        atomicAdd(&sh[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) hist2[wo + k] = sh[k];
}

// pass 2, SCATTER: final entries (low bits stripped) at starts[bucket] + the chunk's share (hist2 holds the exclusive
// prefix over chunks by now).  Windows of up to kS2StageWindow buckets -- every chunk of a dense or moderately sparse
// column -- group the chunk by bucket in LDS first and copy each bucket's run out contiguously; wider windows write
// straight from the counters.
static constexpr u32 kS2StageWindow = 3072;
__global__ void __launch_bounds__(1024) msm_s2_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                       const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P,
                                                       u32 *__restrict__ hist2, const u32 *__restrict__ starts, u32 *__restrict__ entries) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 cidx = blockIdx.x, M = bin_start[P.nh];
    const size_t p0 = (size_t)cidx * P.K2;
    if (p0 >= M) return;
    const u32 p1 = (u32)min((size_t)M, p0 + P.K2);
    const u32 h0 = hlo[cidx], wo = woff[cidx], wsize = woff[cidx + 1] - wo, nbins = wsize >> P.lowb;
    const u32 lowmask = (1u << P.lowb) - 1, strip = P.side ? ~0u : ~(lowmask << P.lb);
    if (wsize > P.lds_window) {             // very sparse column: hist2 (exclusive offsets by now) doubles as the cursor
        const u32 *gb = bin_start + h0 + 1;
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            const u32 k = (rel_bin(gb, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask);
            entries[starts[(h0 << P.lowb) + k] + atomicAdd(&hist2[wo + k], 1u)] = e & strip;
        }
        return;
    }
    if (wsize > kS2StageWindow) {
        u32 *bounds = sh + wsize;
        for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) sh[k] = starts[(h0 << P.lowb) + k] + hist2[wo + k];
        for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
        __syncthreads();
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            const u32 pos = atomicAdd(&sh[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
            entries[pos] = e & strip;
        }
        return;
    }
    u32 *gstart = sh;                      // [wsize] where the chunk's run of bucket k starts in `entries`
    u32 *lstart = gstart + wsize;          // [wsize + 1] ... and in the stage
    u32 *cursor = lstart + wsize + 1;      // [wsize]
    u32 *bounds = cursor + wsize;          // [nbins]
    u32 *stage = bounds + nbins;           // [K2] entries grouped by bucket
    uint16_t *kid = reinterpret_cast<uint16_t *>(stage + P.K2);   // [K2] window-relative bucket of each staged entry
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) {
        gstart[k] = starts[(h0 << P.lowb) + k] + hist2[wo + k];
        lstart[k] = 0;
    }
    for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];
        atomicAdd(&lstart[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 L = wave0_excl_scan(lstart, wsize);
        if (threadIdx.x == 0) lstart[wsize] = L;
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) cursor[k] = lstart[k];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];                                 // second read of the chunk comes from L2
        const u32 k = (rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask);
        const u32 pos = atomicAdd(&cursor[k], 1u);
        stage[pos] = e & strip;
        kid[pos] = (uint16_t)k;
    }
    __syncthreads();
    // consecutive lanes copy consecutive staged entries: runs of one bucket leave as contiguous stores
    for (u32 q = threadIdx.x; q < p1 - (u32)p0; q += blockDim.x) {
        const u32 k = kid[q];
        entries[gstart[k] + (q - lstart[k])] = stage[q];
    }
}

// ---- pass 2 in ONE launch: a workgroup per pass-1 bin ----------------------------------------------------------------------
// The chunked pass 2 above cuts the tagged list into 16 K-entry chunks that may straddle bins, so it needs a plan, a count
// pass, a per-bucket prefix over the chunks, a three-kernel scan of all bucket totals and then the scatter: seven launches,
// the tagged list read twice from HBM.  But pass 1 already knows where every bin begins (bin_start), a bin is one contiguous
// run of the tagged list, and for anything but a pathological column it fits in LDS (2^20 scalars, 17-bit windows: 1024 bins
// of ~15 K entries).  So: one workgroup per bin counts its 2^lowb buckets in LDS, scans them -- starts[bucket] = bin_start +
// local prefix, no global scan -- groups the bin by bucket in LDS and writes it out as ONE contiguous copy.  The tagged list is
// read once from HBM (the second read of a bin hits L2), `entries` is written in full lines.  A bin that does not fit (tens of
// thousands of equal scalars) is scattered straight to memory by the same workgroup.
// Counters are bumped with a wave-aggregated form: when every active lane of a wave holds the same key (a column of equal
// scalars) one lane adds the population count instead of 64 lanes serialising on one LDS address.
__device__ __forceinline__ u32 lds_ticket(u32 *ctr, u32 k) {
    const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)k);
    const unsigned long long act = __ballot(1), same = __ballot(k == k0);
    if (same == act) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0u));
        u32 base = 0;
        if (rank == 0) base = atomicAdd(&ctr[k0], (u32)__popcll(act));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        return base + rank;
    }
    return atomicAdd(&ctr[k], 1u);
}
// the same for four keys per lane: when all four are valid and every active lane's four keys are ONE key, a single atomic takes
// 4 x population; else four tickets.  pos[j] is only written for j < nvalid.
static constexpr u32 kMaxBig = 32, kBigChunks = 64;      // big bins sorted by the chunked kernels below; workgroups per big bin
__device__ __forceinline__ void lds_ticket4(u32 *ctr, const u32 k[4], u32 nvalid, u32 pos[4]) {
    const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)k[0]);
    const bool mine = nvalid == 4 && k[0] == k0 && k[1] == k0 && k[2] == k0 && k[3] == k0;
    const unsigned long long act = __ballot(1), same = __ballot(mine);
    if (same == act) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0u));
        u32 base = 0;
        if (rank == 0) base = atomicAdd(&ctr[k0], 4u * (u32)__popcll(act));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base) + 4u * rank;
#pragma unroll
        for (int j = 0; j < 4; ++j) pos[j] = base + j;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((u32)j < nvalid) pos[j] = lds_ticket(ctr, k[j]);
}
__global__ void __launch_bounds__(1024) msm_s2_bins(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                    Sort2 P, u32 total_buckets, u32 cap, u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    u32 *__restrict__ big, u32 max_big, u32 *__restrict__ zero9, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    u32 cb = 0;                                                         // joined columns: where this column's entries begin
    if (gridDim.z > 1) {
        if (zero9) zero9 = H2_COLZ(zero9, cs.buckets);
        cb = col_entry_base(bin_start, cs);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        starts = H2_COLZ(starts, cs.starts);
        entries = H2_COLZ(entries, cs.entries);
        big = H2_COLZ(big, cs.plan);
    }
    const u32 nbk = 1u << P.lowb, h = blockIdx.x;
    u32 *cnt = sh, *cursor = sh + nbk, *stage = cursor + nbk;          // [nbk] | [nbk] | [cap]
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], E = p1 - p0, q0 = cb + p0;     // q0: the bin's place in the sorted list
    const u32 lowmask = nbk - 1, strip = P.side ? ~0u : ~(lowmask << P.lb);
    // M  (the sentinel behind it: msm_s1_prefix).  Joined columns: that slot is bucket 0 of the next column, which writes the same value.
    if (h == gridDim.x - 1 && threadIdx.x == 0) starts[total_buckets] = cb + bin_start[gridDim.x];
    if (zero9) {
        // the raw bucket slots msm_accumulate parks segments in start from zero: this bin's buckets are cleared HERE (a memset node
        // less on the stream of every commit; the slots are not touched again before the accumulate)
        const u32 b0 = h << P.lowb, b1 = min(total_buckets, b0 + nbk);
        if (b1 > b0) {
            uint4 *z = reinterpret_cast<uint4 *>(zero9 + 36 * (size_t)b0);
            for (u32 i = threadIdx.x; i < 9 * (b1 - b0); i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
        }
    }
    if (E > cap && max_big) {
        // a bin that does not fit the stage (a degenerate column: every scalar equal, half of them 1 ...) goes on the list of big
        // bins, which msm_s2_big_* sort with kBigChunks workgroups each; only past kMaxBig such bins does this workgroup do it alone
        if (threadIdx.x == 0) {
            const u32 slot = atomicAdd(&big[0], 1u);
            if (slot < max_big) big[1 + slot] = h;
            cnt[0] = slot;
        }
        __syncthreads();
        const u32 slot = cnt[0];
        __syncthreads();
        if (slot < max_big) return;
    }
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    // four consecutive entries per lane and trip: four loads in flight per lane instead of one (a bin of a degenerate column --
    // every scalar equal, or half of them 1 -- holds up to 2^20 entries and is streamed by this one workgroup, twice)
    const u32 step = blockDim.x * 4;
    for (u32 base = p0 + threadIdx.x * 4; base < p1; base += step) {
        u32 e[4], k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < p1 ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < p1 ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        u32 pos[4];
        lds_ticket4(cnt, k, min(4u, p1 - base), pos);
    }
    __syncthreads();
    if (threadIdx.x < 64) (void)wave0_excl_scan(cnt, nbk);              // cnt[k] = entries of the bin before bucket k
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        const u32 b = (h << P.lowb) + k;
        cursor[k] = cnt[k];
        if (b < total_buckets) starts[b] = q0 + cnt[k];
    }
    __syncthreads();
    const bool fits = E <= cap;
    for (u32 base = p0 + threadIdx.x * 4; base < p1; base += step) {    // second read of the bin: L2 (a bin beyond the stage that found no
                                                                        // slot on the big-bin list: scattered by this workgroup alone)
        u32 e[4], k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < p1 ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < p1 ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        u32 pos[4];
        lds_ticket4(cursor, k, min(4u, p1 - base), pos);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < p1) {
                if (fits) stage[pos[j]] = e[j] & strip;
                else entries[q0 + pos[j]] = e[j] & strip;               // a bin beyond the stage: scattered straight to memory
            }
    }
    if (!fits) return;
    __syncthreads();
    for (u32 i = threadIdx.x; i < E; i += blockDim.x) entries[q0 + i] = stage[i];
}

// ---- big bins (listed by msm_s2_bins): kBigChunks workgroups per bin -- count, prefix, scatter.  Three small launches that
// return at once when the list is empty (the common case: ~14 us), so that a degenerate column costs what it cost with the
// chunked pass 2 instead of being streamed by one workgroup per bin (every scalar equal: 1.68 ms against 0.88).
__global__ void __launch_bounds__(1024) msm_s2_big_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                         Sort2 P, const u32 *__restrict__ big, u32 *__restrict__ gcnt, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 s = blockIdx.y, c = blockIdx.x;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, lowmask = nbk - 1, h = big[1 + s];
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], csize = (p1 - p0 + kBigChunks - 1) / kBigChunks;
    const u32 a = min(p1, p0 + c * csize), b = min(p1, a + csize);
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    for (u32 base = a + threadIdx.x * 4; base < b; base += blockDim.x * 4) {
        u32 e[4], k[4], pos[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < b ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < b ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        lds_ticket4(sh, k, min(4u, b - base), pos);
    }
    __syncthreads();
    u32 *dst = gcnt + ((size_t)s * kBigChunks + c) * nbk;
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) dst[k] = sh[k];
}
__global__ void __launch_bounds__(1024) msm_s2_big_prefix(const u32 *__restrict__ bin_start, Sort2 P, u32 total_buckets, const u32 *__restrict__ big,
                                                          u32 *__restrict__ gcnt, u32 *__restrict__ starts, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];      // [nbk] bucket totals -> exclusive prefix
    const u32 s = blockIdx.x;
    u32 cb = 0;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        cb = col_entry_base(bin_start, cs);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
        starts = H2_COLZ(starts, cs.starts);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, h = big[1 + s], p0 = cb + bin_start[h];
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        u32 run = 0;
        for (u32 c = 0; c < kBigChunks; ++c) {
            u32 *q = gcnt + ((size_t)s * kBigChunks + c) * nbk + k;
            const u32 t = *q;
            *q = run;                                               // this chunk's offset inside bucket k
            run += t;
        }
        sh[k] = run;
    }
    __syncthreads();
    if (threadIdx.x < 64) (void)wave0_excl_scan(sh, nbk);
    __syncthreads();
    u32 *base = gcnt + (size_t)kMaxBig * kBigChunks * nbk + (size_t)s * nbk;       // bucket offsets inside the bin, for the scatter
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        const u32 b = (h << P.lowb) + k;
        base[k] = sh[k];
        if (b < total_buckets) starts[b] = p0 + sh[k];
    }
}
__global__ void __launch_bounds__(1024) msm_s2_big_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                           Sort2 P, const u32 *__restrict__ big, const u32 *__restrict__ gcnt, u32 *__restrict__ entries,
                                                           ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 s = blockIdx.y, c = blockIdx.x;
    u32 cb = 0;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        cb = col_entry_base(bin_start, cs);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
        entries = H2_COLZ(entries, cs.entries);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, lowmask = nbk - 1, strip = P.side ? ~0u : ~(lowmask << P.lb), h = big[1 + s];
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], csize = (p1 - p0 + kBigChunks - 1) / kBigChunks;
    const u32 a = min(p1, p0 + c * csize), b = min(p1, a + csize);
    const u32 *off = gcnt + ((size_t)s * kBigChunks + c) * nbk, *base_k = gcnt + (size_t)kMaxBig * kBigChunks * nbk + (size_t)s * nbk;
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) sh[k] = base_k[k] + off[k];
    __syncthreads();
    for (u32 base = a + threadIdx.x * 4; base < b; base += blockDim.x * 4) {
        u32 e[4], k[4], pos[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < b ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < b ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        lds_ticket4(sh, k, min(4u, b - base), pos);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < b) entries[cb + p0 + pos[j]] = e[j] & strip;
    }
}

// per-bucket totals; each chunk's count becomes the bucket-relative offset of that chunk
__global__ void __launch_bounds__(256) msm_s2_prefix(u32 *__restrict__ hist2, const u32 *__restrict__ bin_start, const u32 *__restrict__ hlo,
                                                     const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ counts, u32 NB) {
    H2_LATENCY_STAGE();
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= NB) return;
    const u32 h = j >> P.lowb, b0 = bin_start[h], b1 = bin_start[h + 1];
    u32 run = 0;
    if (b1 > b0) {
        const u32 c0 = b0 / P.K2, c1 = (b1 - 1) / P.K2;
        for (u32 cidx = c0; cidx <= c1; ++cidx) {
            const size_t k = (size_t)woff[cidx] + (j - (hlo[cidx] << P.lowb));
            const u32 t = hist2[k];
            hist2[k] = run;
            run += t;
        }
    }
    counts[j] = run;
}

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

// ---- accumulate: exact static partition of the sorted entry list ------------------------------------
// The M sorted entries are cut into T equal ranges, T = the number of lanes the chip keeps resident for
// this kernel, so every lane does the same number of mixed additions and the launch is ONE full round
// (bucket-aligned parts left a 25 % tail: 1.5 rounds of work dispatched as 2).  A lane's range may span
// several buckets: its first segment -- the bucket already open at the range start -- goes to
// heads[t]; every later segment starts a new bucket and is that bucket's only non-head segment, stored
// straight into buckets[b] (zeroed beforehand).  bucket b = buckets[b] + sum of heads[t] for
// ceil(start_b / chunk) <= t < ceil(start_{b+1} / chunk), which msm_finish_buckets adds up.
// GLV: entries index 2m columns; column m + i is phi(P_i) = (zeta x_i, y_i), formed on the fly (extra_index = m then)
// generic path (arbitrary bases + endomorphism split): the n caller-supplied points (reference Montgomery form) are converted
// ONCE per call into M9 form, together with phi(P_i) = (zeta x_i, y_i): column i -> out[i], column n + i -> out[n + i].  The
// bucket accumulation then runs on the carry-free layer exactly as for a registered table, instead of paying a zeta
// multiplication and two form conversions on each of the ~9 entries that read a point.
template <int FB>
__global__ void __launch_bounds__(256) msm_bases_to_m9_glv(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const affine<FB> p = aff_load<FB>(bases + 16 * (size_t)i);
    const affine<FB> q = aff_to_m9<FB>(p);
    affine<FB> phi = q;
    phi.x = fe_mulx<FB>(q.x, glv_zeta<FB>());          // zeta in Montgomery form: (x 2^261)(zeta 2^256) / 2^256
    u32 *d0 = out + 16 * (size_t)i, *d1 = out + 16 * ((size_t)n + i);
    fe_store(d0, q.x);
    fe_store(d0 + 8, q.y);
    fe_store(d1, phi.x);
    fe_store(d1 + 8, phi.y);
}

#ifndef H2_ACC_LOOP
#define H2_ACC_LOOP 2       // 1: the round-3 loop (gather issued before the point is repacked); 2: point consumed first (round 4) -- A/B builds
#endif
#ifndef H2_ACC9_WAVES
#define H2_ACC9_WAVES 2     // waves per SIMD the M9 accumulate is compiled for: 2, 3 and 4 run the adds equally fast (profiles/r02_ubench_fe9.txt);
                            // at 2 the register file keeps room for the sort / fold kernels of commits on other streams (3 streams: 903 vs 861 M/s)
#endif
// M9: the points come from a registered table, stored in M9 form (x * 2^261 mod p, field9.cuh): the additions run on the
// carry-free 9 x 29-bit field layer (curve9.cuh, 17.7-18.0 G mixed adds/s against 13.9-15.0 for the 8 x 32 layer,
// profiles/r02_ubench_fe9.txt) and a flushed segment is converted back to the reference's Montgomery form, canonical, so
// everything downstream (finish, fold, combine) is unchanged.
// A 4-byte global load WITH its wait, as one statement the compiler cannot look into: used on the rare path of msm_accumulate
// only.  A load the compiler tracks, issued under a condition and used after the join, makes it wait for EVERYTHING outstanding
// at that join on every path (vmcnt counts in order) -- on the common path that would be the gathers issued a moment before.
#ifndef H2_ACC_NT
#define H2_ACC_NT 0         // 1: the table gathers of msm_accumulate carry the non-temporal hint (A/B only: profiles/r04_ab_gather_nt.txt)
#endif
template <int F> __device__ __forceinline__ affine<F> aff_gather(const u32 *p) {
#if H2_ACC_NT
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u *q = reinterpret_cast<const v4u *>(p);
    const v4u a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1), c = __builtin_nontemporal_load(q + 2),
              d = __builtin_nontemporal_load(q + 3);
    return affine<F>{fe{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}}, fe{{c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w}}};
#else
    return aff_load<F>(p);
#endif
}
__device__ __forceinline__ u32 load_u32_waited(const u32 *p) {
    u32 v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// A/B only (profiles/r05_acc_power_vs_traffic.txt): -DH2_ACC_GATHER_MASK=0x3FFF folds every table index of the registered path's
// accumulate into the table's first 16384 points (1 MiB: resident in every XCD's L2) -- the same instruction stream with ~no HBM
// traffic, WRONG results (the native driver's parity line fails by design): does the 2.4 GB per launch of half-used gather lines
// cost shader clock under the socket's power limit?
#ifndef H2_ACC_GATHER_MASK
#define H2_ACC_GATHER_MASK 0x7FFFFFFFu
#endif
template <int FB, bool GLV, bool M9 = false>
__global__ void __launch_bounds__(256, (M9 ? H2_ACC9_WAVES : 4)) msm_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (M9 && gridDim.z > 1) {           // column-batched commit: blockIdx.z = column (the table is shared)
        entries = H2_COLZ(entries, cs.entries);
        starts = H2_COLZ(starts, cs.starts);
        heads = H2_COLZ(heads, cs.heads);
        buckets = H2_COLZ(buckets, cs.buckets);
    }
    // `starts` may be a VIEW into a longer boundary array (a group of window slices of a generic multiexp accumulated on its own,
    // msm_launch's slice split): its first entry is then the group's offset into `entries`, not zero; the T ranges tile [base, base + M)
    const u32 base = starts[0];
    const u32 M = starts[total_buckets] - base;
    T = eff_lanes(M, T, div);
    if (t >= T) return;
    const u32 chunk = (M + T - 1) / T;
    const u32 lo = base + min(M, t * chunk), hi = min(base + M, lo + chunk);
    if (M9) {
        xyzz9<FB> acc = xyzz9_identity<FB>();
        if (lo < hi) {
            u32 b = upper_bucket(starts, total_buckets, lo);
            u32 bend = starts[b + 1];
            // Memory operations and the wave's wait counter.  vmcnt counts loads AND stores in issue order, and at a point that
            // some lanes' control flow reaches with conditional operations in flight the compiler has to wait for ALL of them
            // (vmcnt(0)).  The loop is therefore arranged so that nothing young is ever outstanding where a wait falls:
            //   * the gather of the next point and the read of entry i + 2 are UNCONDITIONAL (clamped at the tail), issued
            //     right after the point gathered one addition ago has been consumed (the asm pin below is that point);
            //   * the flush of a bucket boundary -- nine stores and the read of the boundary after next -- is DEFERRED to the
            //     top of the following iteration, behind the gathers, so that a whole mixed addition (~4 us) passes before the
            //     next wait.  At 17-bit windows a wave crosses a boundary in a quarter of its iterations (240 entries per
            //     bucket, 64 lanes); issued at the bottom of the loop, each one stalled the wave for a memory round trip.
            //   * the END of the bucket after the open one is read at the top of EVERY iteration (one dword, a cache hit) and used
            //     by a flush one iteration later at the earliest: unconditional, so the compiler's wait for it is exact.
            u32 t2_last = starts[b + 2];                 // starts[total_buckets + 1] is a sentinel (msm_scan_apply)
            bool first = true, pending = false, t2_ok = true;
            u32 e0 = entries[lo], e1 = entries[min(lo + 1, hi - 1)];
            affine<FB> nxt = aff_gather<FB>(bases + 16 * (size_t)(e0 & H2_ACC_GATHER_MASK));
            for (u32 i = lo; i < hi; ++i) {
                // everything issued during the previous iteration -- the gather of this point, entry i + 1, the boundary read, a
                // flush's stores -- has had a whole mixed addition to complete: this wait is free
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if H2_ACC_LOOP == 2
                // The gathered point is CONSUMED (identity test, repacking into nine limbs per coordinate) before the next gather is
                // issued into the same sixteen registers: the packed point's live range ends at the pin below, so there is no
                // register rotation at the loop's back edge (eleven moves per addition in the form that issued the gather first).
                // The gather still has a whole addition (~7500 cycles) to land.
                const bool p_ident = aff_is_identity(nxt);
                aff9<FB> q = aff9_unpack<FB>(nxt);
                asm volatile("" : "+v"(q.x.v[0]), "+v"(q.x.v[1]), "+v"(q.x.v[2]), "+v"(q.x.v[3]), "+v"(q.x.v[4]), "+v"(q.x.v[5]), "+v"(q.x.v[6]),
                             "+v"(q.x.v[7]), "+v"(q.x.v[8]), "+v"(q.y.v[0]), "+v"(q.y.v[1]), "+v"(q.y.v[2]), "+v"(q.y.v[3]), "+v"(q.y.v[4]),
                             "+v"(q.y.v[5]), "+v"(q.y.v[6]), "+v"(q.y.v[7]), "+v"(q.y.v[8])
                             :
                             : "memory");
#else
                affine<FB> p = nxt;
                asm volatile("" : "+v"(p.x.v[0]), "+v"(p.x.v[1]), "+v"(p.x.v[2]), "+v"(p.x.v[3]), "+v"(p.x.v[4]), "+v"(p.x.v[5]), "+v"(p.x.v[6]),
                             "+v"(p.x.v[7]), "+v"(p.y.v[0]), "+v"(p.y.v[1]), "+v"(p.y.v[2]), "+v"(p.y.v[3]), "+v"(p.y.v[4]), "+v"(p.y.v[5]),
                             "+v"(p.y.v[6]), "+v"(p.y.v[7])
                             :
                             : "memory");
#endif
                const u32 neg = e0 >> 31;
                const u32 e2 = entries[min(i + 2, hi - 1)];
                nxt = aff_gather<FB>(bases + 16 * (size_t)(e1 & H2_ACC_GATHER_MASK));      // at the tail: a stale, valid entry
                const u32 t2_cur = starts[b + 2];
                e0 = e1;
                e1 = e2;
                bool flushed = false;
                if (pending) {
                    // parked as raw limbs (a few stores): the conversion back to the reference's Montgomery form costs most of a
                    // mixed addition and would be paid by the whole wave each time one of its lanes crosses a bucket boundary;
                    // msm_segments_to_r256 does it for all segments at once
                    xyzz9_store_raw<FB>(first ? heads + 36 * (size_t)t : buckets + 36 * (size_t)b, acc);
                    first = false;
                    acc = xyzz9_identity<FB>();
                    ++b;
                    if (t2_ok && t2_last > i) {
                        bend = t2_last;                                              // = starts[b + 1], read an iteration ago
                    } else {
                        // rare: empty buckets follow (sparse columns), or the bucket just closed held a single entry
                        b = upper_bucket(starts, total_buckets, i);
                        bend = load_u32_waited(starts + b + 1);
                    }
                    flushed = true;
                }
#if H2_ACC_LOOP == 2
                if (!p_ident) {
                    if (neg) q.y = fe9_sub(fe9_zero(), q.y);          // signed limbs: negation is nine subtractions
                    xyzz9_madd<FB, true>(acc, q);
                }
#else
                if (!aff_is_identity(p)) {
                    aff9<FB> q = aff9_unpack<FB>(p);
                    if (neg) q.y = fe9_sub(fe9_zero(), q.y);          // signed limbs: negation is nine subtractions
                    xyzz9_madd<FB>(acc, q);
                }
#endif
                pending = i + 1 == bend && i + 1 < hi;
                t2_ok = !flushed;            // the read at the top of an iteration that flushed was made for the bucket it closed
                t2_last = t2_cur;
            }
            xyzz9_store_raw<FB>(first ? heads + 36 * (size_t)t : buckets + 36 * (size_t)b, acc);
            return;
        }
        xyzz9_store_raw<FB>(heads + 36 * (size_t)t, acc);
        return;
    }
    xyzz<FB> acc = xyzz_identity<FB>();
    if (lo < hi) {
        u32 b = upper_bucket(starts, total_buckets, lo);
        u32 bend = starts[b + 1];
        bool first = true;
        u32 e0 = entries[lo], e1 = lo + 1 < hi ? entries[lo + 1] : 0;
        u32 idx = e0 & 0x7FFFFFFFu;
        bool phi = GLV && idx >= extra_index, phi_nxt = false;
        // the blind's base `w` (Params::commit, poly/commitment.rs:127) may live in its own buffer
        affine<FB> nxt = GLV ? aff_load<FB>(bases + 16 * (size_t)(phi ? idx - extra_index : idx))
                             : aff_load<FB>(idx == extra_index ? extra_base : bases + 16 * (size_t)idx);
        for (u32 i = lo; i < hi; ++i) {
            affine<FB> p = nxt;
            const u32 neg = e0 >> 31;
            // software pipeline: entry i+2 and base i+1 are in flight while point i is added
            const u32 e2 = i + 2 < hi ? entries[i + 2] : 0;
            if (i + 1 < hi) {
                idx = e1 & 0x7FFFFFFFu;
                phi_nxt = GLV && idx >= extra_index;
                nxt = GLV ? aff_load<FB>(bases + 16 * (size_t)(phi_nxt ? idx - extra_index : idx))
                          : aff_load<FB>(idx == extra_index ? extra_base : bases + 16 * (size_t)idx);
            }
            e0 = e1;
            e1 = e2;
            if (GLV && phi) p.x = fe_mulx<FB>(p.x, glv_zeta<FB>());
            phi = phi_nxt;
            if (neg) p.y = fe_neg<FB>(p.y);
            xyzz_madd_lazy<FB>(acc, p);
            if (i + 1 == bend && i + 1 < hi) {  // bucket b ends inside the range: flush, open the next non-empty bucket
                xyzz_reduce_lazy<FB>(acc);
                xyzz_store<FB>(first ? heads + 32 * (size_t)t : buckets + 32 * (size_t)b, acc);
                first = false;
                acc = xyzz_identity<FB>();
                do { ++b; bend = starts[b + 1]; } while (bend <= i + 1);
            }
        }
        xyzz_reduce_lazy<FB>(acc);
        xyzz_store<FB>(first ? heads + 32 * (size_t)t : buckets + 32 * (size_t)b, acc);
        if (first) return;
        acc = xyzz_identity<FB>();
    }
    if (lo >= hi) xyzz_store<FB>(heads + 32 * (size_t)t, acc);
}

// raw M9 segments (heads of the T ranges, then the bucket slots; 36 words each, untouched bucket slots are zero) ->
// XYZZ in the reference's Montgomery form, canonical, 32 words each: what the finisher and the fold read
template <int FB>
__global__ void __launch_bounds__(256) msm_segments_to_r256(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets) {
    H2_LATENCY_STAGE();
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= T + total_buckets) return;
    const xyzz9<FB> a = xyzz9_load_raw<FB>(raw + 36 * (size_t)s);
    u32 *dst = s < T ? heads + 32 * (size_t)s : buckets + 32 * (size_t)(s - T);
    xyzz_store<FB>(dst, xyzz9_is_identity(a) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(a));
}

// ---- finisher: bucket b = its own non-head segment + the heads of the ranges that begin inside it -----
// Buckets owning more than kHeavy heads (scalars repeated thousands of times) are parked on a list and summed
// by a whole workgroup each in msm_finish_heavy.
static constexpr u32 kHeavy = 64;
static constexpr u32 kMaxHeavy = 512;   // heavy buckets handed to the workgroup path; any beyond that are summed in place
template <int FB>
__global__ void __launch_bounds__(256) msm_finish_buckets(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div) {
    H2_LATENCY_STAGE();
    // one quad of lanes per bucket (curve_wide.cuh)
    const u32 b = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (b >= total_buckets) return;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 M = starts[total_buckets];
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    const u32 h0 = (starts[b] + chunk - 1) / chunk, h1 = (starts[b + 1] + chunk - 1) / chunk;
    if (h1 <= h0) return;
    if (h1 - h0 > kHeavy) {
        u32 slot = 0;
        if (lead) slot = atomicAdd(&heavy[1], 1u);
        slot = (u32)__builtin_amdgcn_mov_dpp((int)slot, 0, 0xf, 0xf, false);   // quad lane 0's ticket
        if (slot < kMaxHeavy) {
            if (lead) {
                heavy[2 + slot] = b;
                atomicAdd(&heavy[0], 1u);
            }
            return;
        }
    }
    xyzz<FB> acc = xyzz_load<FB>(buckets + 32 * (size_t)b);
    xyzz<FB> nxt = xyzz_load<FB>(heads + 32 * (size_t)h0);
    for (u32 t = h0; t < h1; ++t) {
        xyzz<FB> p = nxt;
        if (t + 1 < h1) nxt = xyzz_load<FB>(heads + 32 * (size_t)(t + 1));
        xyzz_add_wide<FB>(acc, p);
    }
    if (lead) xyzz_store<FB>(buckets + 32 * (size_t)b, acc);
}
// heavy buckets: kHeavyBlocks workgroups share one bucket's heads (quad-wide adds), then msm_finish_heavy2 folds
// their partial sums and the bucket's own segment
static constexpr u32 kHeavyBlocks = 32;
static constexpr u32 kHeavyRows = 16;     // workgroup rows of the heavy-bucket launches: they walk the list (at most kMaxHeavy long, usually empty)
template <int FB>
__global__ void __launch_bounds__(256) msm_finish_heavy(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    if (blockIdx.y >= min(heavy[1], kMaxHeavy)) return;
    const u32 b = heavy[2 + blockIdx.y], t = threadIdx.x / kGroup, nl = blockDim.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 M = starts[total_buckets];
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    const u32 h0 = (starts[b] + chunk - 1) / chunk, h1 = (starts[b + 1] + chunk - 1) / chunk;
    const u32 share = (h1 - h0 + kHeavyBlocks - 1) / kHeavyBlocks;
    const u32 lo = h0 + blockIdx.x * share, hi = min(h1, lo + share);
    xyzz<FB> acc = xyzz_identity<FB>();
    for (u32 i = lo + t; i < hi; i += nl) {
        xyzz<FB> p = xyzz_load<FB>(heads + 32 * (size_t)i);
        xyzz_add_wide<FB>(acc, p);
    }
    if (lead) xyzz_store<FB>(sh + 32 * t, acc);
    __syncthreads();
    for (u32 off = nl / 2; off > 0; off >>= 1) {
        if (t < off) {
            xyzz<FB> x = xyzz_load<FB>(sh + 32 * t), y = xyzz_load<FB>(sh + 32 * (t + off));
            xyzz_add_wide<FB>(x, y);
            if (lead) xyzz_store<FB>(sh + 32 * t, x);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        xyzz<FB> r = xyzz_load<FB>(sh);
        xyzz_store<FB>(scratch + 32 * ((size_t)blockIdx.y * kHeavyBlocks + blockIdx.x), r);
    }
}
template <int FB>
__global__ void __launch_bounds__(64) msm_finish_heavy2(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 16];
    if (blockIdx.x >= min(heavy[1], kMaxHeavy)) return;
    // one wave = 16 quads: quad q adds partials q and q + 16, then a 4-level tree (5 dependent additions instead of 32)
    const u32 b = heavy[2 + blockIdx.x], q = threadIdx.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 *src = scratch + 32 * ((size_t)blockIdx.x * kHeavyBlocks);
    xyzz<FB> acc = xyzz_load<FB>(src + 32 * q);
    for (u32 i = q + 16; i < kHeavyBlocks; i += 16) {
        xyzz<FB> p = xyzz_load<FB>(src + 32 * i);
        xyzz_add_wide<FB>(acc, p);
    }
    if (lead) xyzz_store<FB>(sh + 32 * q, acc);
    __syncthreads();
    for (u32 off = 8; off > 0; off >>= 1) {
        if (q < off) {
            xyzz<FB> x = xyzz_load<FB>(sh + 32 * q), y = xyzz_load<FB>(sh + 32 * (q + off));
            xyzz_add_wide<FB>(x, y);
            if (lead) xyzz_store<FB>(sh + 32 * q, x);
        }
        __syncthreads();
    }
    if (q == 0) {
        xyzz<FB> r = xyzz_load<FB>(sh), own = xyzz_load<FB>(buckets + 32 * (size_t)b);
        xyzz_add_wide<FB>(r, own);
        if (lead) xyzz_store<FB>(buckets + 32 * (size_t)b, r);
    }
}

// total[b] += part[b] over one bucket slice (the ranges of a commit assembled from chunks share one fold): one quad per bucket
template <int FB>
__global__ void __launch_bounds__(256) msm_bucket_add(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb) {
    H2_LATENCY_STAGE();
    const u32 b = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (b >= nb) return;
    const xyzz<FB> p = xyzz_load<FB>(part + 32 * (size_t)b);
    if (fe_is_zero(p.zz)) return;                              // an empty bucket of this range
    xyzz<FB> acc = xyzz_load<FB>(total + 32 * (size_t)b);
    xyzz_add_wide<FB>(acc, p);
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz_store<FB>(total + 32 * (size_t)b, acc);
}

// ---- the first levels of the fold of a WIDE bucket slice (registered tables from 2^16 buckets), throughput form ----------------
// The quad-lane kernels below spend four lanes on a point operation to cut its latency; on the two stages that touch EVERY
// bucket -- finishing (bucket += heads) and the row / column sums -- that is 2.4x the VALU work of a one-lane addition on the
// carry-free layer, at 4096 waves, right when another stream's msm_accumulate wants the SIMDs.  These two stages therefore run
// one lane per point on the raw M9 segments msm_accumulate parks (no msm_segments_to_r256 pass), and only the S + NR row /
// column sums are converted for the latency-bound tail (msm_reduce_segments on a few hundred points).
template <int FB>
__global__ void __launch_bounds__(256) fold9_finish(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs) {
    H2_LATENCY_STAGE();
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= total_buckets) return;
    if (gridDim.z > 1) {
        heads9 = H2_COLZ(heads9, cs.heads);
        starts = H2_COLZ(starts, cs.starts);
        buckets9 = H2_COLZ(buckets9, cs.buckets);
        heavy = H2_COLZ(heavy, cs.heavy);
    }
    const u32 base = starts[0];                      // (a slice group's view: see msm_accumulate)
    const u32 M = starts[total_buckets] - base;
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    const u32 h0 = (starts[b] - base + chunk - 1) / chunk, h1 = (starts[b + 1] - base + chunk - 1) / chunk;
    if (h1 <= h0) return;
    if (h1 - h0 > kHeavy) {
        const u32 slot = atomicAdd(&heavy[1], 1u);
        if (slot < kMaxHeavy) {
            heavy[2 + slot] = b;
            atomicAdd(&heavy[0], 1u);
            return;
        }
    }
    xyzz9<FB> acc = xyzz9_load_raw<FB>(buckets9 + 36 * (size_t)b);
    for (u32 t = h0; t < h1; ++t) xyzz9_add<FB>(acc, xyzz9_load_raw<FB>(heads9 + 36 * (size_t)t));
    xyzz9_store_raw<FB>(buckets9 + 36 * (size_t)b, acc);
}
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
// heavy buckets (more than kHeavy heads): kHeavyBlocks workgroups share the heads, fold9_finish_heavy2 adds their sums to the bucket
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_finish_heavy(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    if (gridDim.z > 1) {
        heavy = H2_COLZ(heavy, cs.heavy);
        heads9 = H2_COLZ(heads9, cs.heads);
        starts = H2_COLZ(starts, cs.starts);
        scratch9 = H2_COLZ(scratch9, cs.hscratch);
    }
    // gridDim.y = kHeavyRows workgroup rows walk the list of heavy buckets (round 5: the launch used to carry one row per POSSIBLE heavy
    // bucket -- 32 x 512 workgroups that found an empty list and left, ~12 us of dispatch per commit; a column has no heavy bucket
    // unless it is degenerate, and then a handful)
    const u32 count = min(heavy[1], kMaxHeavy);
    const u32 base = starts[0];
    const u32 M = starts[total_buckets] - base;
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    for (u32 slot = blockIdx.y; slot < count; slot += gridDim.y) {
        const u32 b = heavy[2 + slot];
        const u32 h0 = (starts[b] - base + chunk - 1) / chunk, h1 = (starts[b + 1] - base + chunk - 1) / chunk;
        const u32 share = (h1 - h0 + kHeavyBlocks - 1) / kHeavyBlocks;
        const u32 lo = h0 + blockIdx.x * share, hi = min(h1, lo + share);
        xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(heads9, hi > lo ? hi - lo : 0u, [lo](u32 k) { return lo + k; });
        acc = fold9_quads_sum<FB>(acc, sh);
        if (fold9_root() && (threadIdx.x & (kGroup - 1)) == 0) xyzz9_store_raw<FB>(scratch9 + 36 * ((size_t)slot * kHeavyBlocks + blockIdx.x), acc);
        __syncthreads();                                 // `sh` is reused by the next bucket's tree
    }
}
template <int FB>
__global__ void __launch_bounds__(64, 3) fold9_finish_heavy2(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[8 * 36];
    if (gridDim.z > 1) {
        heavy = H2_COLZ(heavy, cs.heavy);
        scratch9 = H2_COLZ(scratch9, cs.hscratch);
        buckets9 = H2_COLZ(buckets9, cs.buckets);
    }
    const u32 count = min(heavy[1], kMaxHeavy);
    for (u32 slot = blockIdx.x; slot < count; slot += gridDim.x) {          // (kHeavyRows workgroups walk the list: see fold9_finish_heavy)
        // 16 quads: two partials each, a 4-level tree, then the bucket's own segment (6 dependent additions)
        const u32 b = heavy[2 + slot];
        const u32 *src = scratch9 + 36 * ((size_t)slot * kHeavyBlocks);
        xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(src, kHeavyBlocks, [](u32 k) { return k; });
        acc = fold9_quads_sum<FB>(acc, sh);
        if (fold9_root()) {
            xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(buckets9 + 36 * (size_t)b));
            if ((threadIdx.x & (kGroup - 1)) == 0) xyzz9_store_raw<FB>(buckets9 + 36 * (size_t)b, acc);
        }
        __syncthreads();
    }
}
// row / column sums of the NR x S bucket matrix (see msm_rowcol_sums for the algebra): one workgroup of 64 quads per line,
// lines9[lo] = C_lo (lo < S), lines9[S + hi] = R_hi (1 <= hi < NR; row 0 carries weight 0 and is never formed), raw M9.
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_rowcol(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    if (gridDim.z > 1) {
        buckets9 = H2_COLZ(buckets9, cs.buckets);
        lines9 = H2_COLZ(lines9, cs.lines);
    }
    const bool is_col = blockIdx.x < S;
    const u32 id = is_col ? blockIdx.x : blockIdx.x - S + 1;          // column lo, or row hi
    const u32 cnt = is_col ? NR : S;
    const u32 base = blockIdx.y * S * NR + (is_col ? id : id * S), step = is_col ? S : 1;      // blockIdx.y: the bucket slice (paired commits: 2)
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(buckets9, cnt, [base, step](u32 k) { return base + k * step; });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (fold9_root() && (threadIdx.x & (kGroup - 1)) == 0)
        xyzz9_store_raw<FB>(lines9 + 36 * ((size_t)blockIdx.y * (S + NR) + (is_col ? id : S + id)), acc);
}
// The rest of the fold of a wide slice in ONE launch.  sum_j (j + 1) B_j = sum_lo (lo + 1) C_lo + S sum_hi hi R_hi is a sum of
// `planes` = log2 S + log2 NR bit planes: plane t (weight 2^t) holds the columns with bit t of lo + 1 set (t < log2 S; plane
// log2 S holds C_{S-1} alone) and the rows with bit t - log2 S of hi set.  Workgroup t sums plane t -- a tree over at most
// S / 2 + NR / 2 lines --, doubles the sum t times (the planes do that side by side: the top plane's t doublings are the chain
// nothing shortens, everything else hides behind it), and the workgroup that finishes LAST (a counter behind a fence; it leaves
// the counter at zero for the next launch) adds the 16 weighted planes with one more tree: ~8 + 15 + 5 dependent quad-lane
// operations, against the ~60 of a running sum over segments, a slice tree and a Horner step in three launches
// (reduce_segments + sum_slice + combine: 126 us of a 1.28 ms commit; this kernel: ~70).
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_planes(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    __shared__ u32 s_last;
    if (gridDim.z > 1) {
        lines9 = H2_COLZ(lines9, cs.lines);
        planes9 = H2_COLZ(planes9, cs.planes);
        counter = H2_COLZ(counter, cs.ctr);
        out = co.out[blockIdx.z];
    }
    const u32 t = blockIdx.x, planes = gridDim.x;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    lines9 += 36 * (size_t)blockIdx.y * (S + NR);                     // blockIdx.y: the bucket slice = the output (paired commits: 2)
    planes9 += 36 * (size_t)blockIdx.y * 32;
    counter += blockIdx.y;
    out += (out_kind == kOutSliceSum ? 32 : out_kind == H2_OUT_AFFINE ? 16 : 24) * (size_t)blockIdx.y;
    const u32 ncol = (int)t < cb ? S / 2 : (int)t == cb ? 1u : 0u, nrow = (int)t >= cb ? NR / 2 : 0u;
    const u32 jc = t, jr = t - (u32)cb;
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(lines9, ncol + nrow, [=](u32 k) {
        if (k < ncol) {
            if ((int)jc == cb) return S - 1;                                                         // lo + 1 = S
            return ((((k >> jc) << (jc + 1)) | (1u << jc) | (k & ((1u << jc) - 1u))) - 1u);         // k-th value of lo + 1 with bit jc set
        }
        const u32 r = k - ncol;
        return S + (((r >> jr) << (jr + 1)) | (1u << jr) | (r & ((1u << jr) - 1u)));                // k-th hi with bit jr set
    });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (fold9_root()) {
        for (u32 k = 0; k < t; ++k) acc = xyzz9_dbl_wide<FB>(acc);         // the plane's weight, applied here: the planes double side by side
        if (lead) {
            xyzz9_store_raw<FB>(planes9 + 36 * (size_t)t, acc);
            __threadfence();
            s_last = atomicAdd(counter, 1u) == planes - 1 ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // the workgroup that arrived last adds the weighted planes (a tree again)
    acc = fold9_quad_gather<FB, H2_FOLD_D>(planes9, planes, [](u32 k) { return k; });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (!fold9_root()) return;
    const xyzz<FB> r = xyzz9_to_r256_wide<FB>(acc);
    if (!lead) return;
    *counter = 0;
    if (out_kind == kOutSliceSum) {              // a window slice of a generic multiexp: XYZZ in the reference's form, for msm_combine
        xyzz_store<FB>(out, r);
        return;
    }
    if (out_kind == H2_OUT_AFFINE) {
        affine<FB> o = xyzz_to_affine<FB>(r);
        if (!out_mont) { o.x = fe_from_mont<FB>(o.x); o.y = fe_from_mont<FB>(o.y); }
        fe_store(out, o.x);
        fe_store(out + 8, o.y);
    } else {
        fe X, Y, Z;
        xyzz_to_jacobian<FB>(r, X, Y, Z);
        if (!out_mont) { X = fe_from_mont<FB>(X); Y = fe_from_mont<FB>(Y); Z = fe_from_mont<FB>(Z); }
        fe_store(out, X);
        fe_store(out + 8, Y);
        fe_store(out + 16, Z);
    }
}

// The three tail kernels below run each logical lane as a quad of 4 hardware lanes (curve_wide.cuh): the chip
// is nearly idle here, so lanes are free and the dependent-multiply depth per point operation drops 3x.

// ---- reduce level 1: segment of kSeg buckets -> sum_j (j+1) * B_j restricted to the segment ------
template <int FB>
__global__ void __launch_bounds__(256) msm_reduce_segments(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg) {
    H2_LATENCY_STAGE();
    const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (t >= total_segments) return;
    u32 segs_per_slice = NB / seg;
    u32 sl = t / segs_per_slice, sg = t % segs_per_slice;
    const u32 *base = buckets + 32 * ((size_t)sl * NB + (size_t)sg * seg);
    xyzz<FB> run = xyzz_identity<FB>(), acc = xyzz_identity<FB>();
    for (int j = seg - 1; j >= 0; --j) {
        xyzz<FB> bk = xyzz_load<FB>(base + 32 * j);
        xyzz_add_wide<FB>(run, bk);
        xyzz_add_wide<FB>(acc, run);
    }
    // buckets of this segment carry weights sg*seg + (j+1): add (sg*seg) * run
    xyzz<FB> sh = xyzz_mul_small_wide<FB>(run, sg * seg);
    xyzz_add_wide<FB>(acc, sh);
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz_store<FB>(partial + 32 * (size_t)t, acc);
}

// ---- reduce level 2: tree sum of a slice's partials, in two launches (many workgroups, then one per slice) --------
// grid (blocks_per_slice, slices); block j of slice s sums partial[s][j * share .. (j + 1) * share) into out[s][j]
template <int FB>
__global__ void __launch_bounds__(256) msm_sum_slice(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 sl = blockIdx.y, blk = blockIdx.x, t = threadIdx.x / kGroup, nl = blockDim.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 lo = blk * share, hi = min(per_slice, lo + share);
    const u32 *src = partial + 32 * (size_t)sl * per_slice;
    xyzz<FB> acc = xyzz_identity<FB>();
    for (u32 i = lo + t; i < hi; i += nl) {
        xyzz<FB> p = xyzz_load<FB>(src + 32 * (size_t)i);
        xyzz_add_wide<FB>(acc, p);
    }
    if (lead) xyzz_store<FB>(sh + 32 * t, acc);
    __syncthreads();
    for (u32 off = nl / 2; off > 0; off >>= 1) {
        if (t < off) {
            xyzz<FB> a = xyzz_load<FB>(sh + 32 * t), b = xyzz_load<FB>(sh + 32 * (t + off));
            xyzz_add_wide<FB>(a, b);
            if (lead) xyzz_store<FB>(sh + 32 * t, a);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        xyzz<FB> r = xyzz_load<FB>(sh);
        xyzz_store<FB>(out + 32 * ((size_t)sl * gridDim.x + blk), r);
    }
}

// ---- wide bucket slices (NB > 2^15, registered path with c > 16): sum_j (j + 1) B_j with j = hi * S + lo splits into
//      S * sum_hi hi * R_hi + sum_lo (lo + 1) * C_lo   (R = row sums, C = column sums of the NR x S bucket matrix),
// i.e. ~one add per bucket, all of them independent (a tree per row / column) instead of a running sum plus a
// small-scalar multiple per 8-bucket segment.  Output laid out as two slices of NR points for the ordinary reduce:
// slice 0 = C_0 .. C_{S-1} (then identities), slice 1 = R_1 .. R_{NR-1} (then one identity); msm_combine's Horner
// step with "window width" log2 S then forms S * (slice 1) + (slice 0).
template <int FB>
__global__ void __launch_bounds__(256) msm_rowcol_sums(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 t = threadIdx.x / kGroup, nl = blockDim.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const bool is_col = blockIdx.x < S;
    const u32 id = is_col ? blockIdx.x : blockIdx.x - S + 1;          // column lo, or row hi (row 0 carries weight 0)
    const u32 cnt = is_col ? NR : S;
    const size_t base = is_col ? id : (size_t)id * S, step = is_col ? S : 1;
    xyzz<FB> acc = xyzz_identity<FB>();
    for (u32 i = t; i < cnt; i += nl) {
        xyzz<FB> p = xyzz_load<FB>(buckets + 32 * (base + (size_t)i * step));
        xyzz_add_wide<FB>(acc, p);
    }
    if (lead) xyzz_store<FB>(sh + 32 * t, acc);
    __syncthreads();
    for (u32 off = nl / 2; off > 0; off >>= 1) {
        if (t < off) {
            xyzz<FB> a = xyzz_load<FB>(sh + 32 * t), b = xyzz_load<FB>(sh + 32 * (t + off));
            xyzz_add_wide<FB>(a, b);
            if (lead) xyzz_store<FB>(sh + 32 * t, a);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        xyzz<FB> r = xyzz_load<FB>(sh);
        xyzz_store<FB>(wide + 32 * (is_col ? (size_t)id : (size_t)NR + id - 1), r);
    }
}

// ---- combine: Horner over slices (windows), emit Jacobian / affine; one quad of lanes ---------------
// extra_dbl / addend / out_kind == kOutSliceSum serve the slice split of a large generic multiexp (msm_launch): the UPPER group of
// slices is summed by Horner, doubled extra_dbl = c x (slices below it) more times and left as XYZZ (32 words); the lower group's
// call then adds that point (`addend`) to its own Horner sum and emits the result.
template <int FB>
__global__ void __launch_bounds__(64) msm_combine(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl = 0, const u32 *__restrict__ addend = nullptr) {
    H2_LATENCY_STAGE();
    if (threadIdx.x >= kGroup) return;
    // one block: Horner over the slices.  Several blocks (pair commits): block b emits slice b alone as output b.
    if (gridDim.x > 1) {
        slice_sums += 32 * (size_t)blockIdx.x;
        out += (out_kind == H2_OUT_AFFINE ? 16 : 24) * (size_t)blockIdx.x;
        slices = 1;
    }
    // the chain runs on the carry-free layer (curve9_wide.cuh): ~850 instructions per doubling against ~1300, and the 128-130
    // doublings of a generic multiexp's Horner step ARE this kernel (355 us of a 0.6-0.7 ms small multiexp before)
    xyzz9<FB> r9 = xyzz9_identity<FB>();
    for (int w = slices - 1; w >= 0; --w) {
        if (w != slices - 1)
            for (int k = 0; k < c; ++k) r9 = xyzz9_dbl_wide<FB>(r9);
        const xyzz9<FB> s9 = xyzz9_from_r256_wide<FB>(xyzz_load<FB>(slice_sums + 32 * (size_t)w));
        xyzz9_add_wide<FB>(r9, s9);
    }
    for (int k = 0; k < extra_dbl; ++k) r9 = xyzz9_dbl_wide<FB>(r9);
    if (addend) xyzz9_add_wide<FB>(r9, xyzz9_from_r256_wide<FB>(xyzz_load<FB>(addend)));
    const xyzz<FB> r = xyzz9_to_r256_wide<FB>(r9);
    if (threadIdx.x != 0) return;
    if (out_kind == kOutSliceSum) {
        xyzz_store<FB>(out, r);
        return;
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

// The Horner step over the window slices of SEVERAL ranges of one multiexp (h2_msm's range pipeline): slice sums are linear in the
// points, so sum_q Horner(S_q) = Horner(sum_q S_q) -- quad w adds slice w's sums over the ranges (side by side), then quad 0 runs ONE
// chain of (slices - 1) c doublings instead of one chain per range.
struct RangeSums {
    const u32 *p[16];
};
template <int FB>
__global__ void __launch_bounds__(64) msm_combine_ranges(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[16 * 36];
    const int w = threadIdx.x / kGroup;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    if (w < slices) {
        xyzz9<FB> acc = xyzz9_identity<FB>();
        for (int q = 0; q < ranges; ++q) xyzz9_add_wide<FB>(acc, xyzz9_from_r256_wide<FB>(xyzz_load<FB>(rs.p[q] + 32 * (size_t)w)));
        if (lead) xyzz9_store_raw<FB>(sh + 36 * w, acc);
    }
    __syncthreads();
    if (threadIdx.x >= kGroup) return;
    xyzz9<FB> r9 = xyzz9_identity<FB>();
    for (int s = slices - 1; s >= 0; --s) {
        if (s != slices - 1)
            for (int k = 0; k < c; ++k) r9 = xyzz9_dbl_wide<FB>(r9);
        xyzz9_add_wide<FB>(r9, xyzz9_load_raw<FB>(sh + 36 * s));
    }
    const xyzz<FB> r = xyzz9_to_r256_wide<FB>(r9);
    if (threadIdx.x != 0) return;
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

// ---- the collapsed generators of the opening argument, straight from a registered table ------------------------------------
// After J rounds G'_J[i] = sum_{h < 2^J} s(h) * G[i + h * nJ] (nJ = 2^(k-J); s(h) = the challenge products of
// h2_ipa_round_scalars_device), i.e. nJ multiexps of 2^J terms that SHARE their scalars.  Each s(h) is cut into the table's
// 16-bit signed digits d_w and every d_w into four signed 4-bit digits e_v in [-7, 8]:
//     s(h) G[m] = sum_w sum_v 16^v e_{h,w,v} T[w][m],          T[w][m] = 2^(16w) G[m] (the table's row w)
// so bucket (v, b) of output i collects +-T[w][i + h nJ] over the (h, w) with |e_{h,w,v}| = b -- the SAME (h, w, sign) list for
// every i.  The host writes the 32 lists once (2^J * 64 entries in all); lane i of workgroup row (v, b) walks list (v, b) with
// no divergence and perfectly coalesced 64-byte gathers (consecutive lanes read consecutive table columns), 2^J * 64 mixed
// additions per output in the carry-free field layer.  ipa_collapse_windows then forms sum_b b * bucket per (i, v) by running
// sums and ipa_collapse_finish the Horner step over v (12 doublings) and the affine result.
template <int FB>
__global__ void __launch_bounds__(256, H2_ACC9_WAVES) ipa_collapse_buckets(const u32 *__restrict__ table, const u32 *__restrict__ list,
                                                                         const u32 *__restrict__ list_start, u32 nJ, u32 *__restrict__ sums) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lb = blockIdx.y;
    if (i >= nJ) return;
    const u32 lo = list_start[lb], hi = list_start[lb + 1];
    xyzz9<FB> acc = xyzz9_identity<FB>();
    if (lo < hi) {
        // loads unconditional (clamped at the tail), as in msm_accumulate: a conditional load makes hipcc wait for everything in flight
        u32 e0 = list[lo], e1 = list[min(lo + 1, hi - 1)];
        affine<FB> nxt = aff_load<FB>(table + 16 * ((size_t)(e0 & 0x7FFFFFFFu) + i));
        for (u32 t = lo; t < hi; ++t) {
            const affine<FB> p = nxt;
            const u32 neg = e0 >> 31;
            const u32 e2 = list[min(t + 2, hi - 1)];
            nxt = aff_load<FB>(table + 16 * ((size_t)(e1 & 0x7FFFFFFFu) + i));
            e0 = e1;
            e1 = e2;
            if (!aff_is_identity(p)) {
                aff9<FB> q = aff9_unpack<FB>(p);
                if (neg) q.y = fe9_sub(fe9_zero(), q.y);
                xyzz9_madd<FB>(acc, q);
            }
        }
    }
    xyzz_store<FB>(sums + 32 * ((size_t)lb * nJ + i), xyzz9_is_identity(acc) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(acc));
}

// lane (i, v): sums[v * 8][i] <- sum_{b = 1..8} b * sums[v * 8 + b - 1][i]
template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse_windows(u32 *__restrict__ sums, u32 nJ) {
    H2_LATENCY_STAGE();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * nJ) return;
    const u32 v = t / nJ, i = t % nJ;
    xyzz<FB> run = xyzz_identity<FB>(), tot = xyzz_identity<FB>();
    for (int b = 8; b >= 1; --b) {
        xyzz_add<FB>(run, xyzz_load<FB>(sums + 32 * ((size_t)(v * 8 + b - 1) * nJ + i)));
        xyzz_add<FB>(tot, run);
    }
    xyzz_store<FB>(sums + 32 * ((size_t)(v * 8) * nJ + i), tot);
}

template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse_finish(const u32 *__restrict__ sums, u32 nJ, u32 *__restrict__ out_xy) {
    H2_LATENCY_STAGE();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nJ) return;
    xyzz<FB> acc = xyzz_load<FB>(sums + 32 * ((size_t)24 * nJ + i));
    for (int v = 2; v >= 0; --v) {
        for (int d = 0; d < 4; ++d) acc = xyzz_dbl<FB>(acc);
        xyzz_add<FB>(acc, xyzz_load<FB>(sums + 32 * ((size_t)(v * 8) * nJ + i)));
    }
    const affine<FB> a = xyzz_to_affine<FB>(acc);
    fe_store(out_xy + 16 * (size_t)i, a.x);
    fe_store(out_xy + 16 * (size_t)i + 8, a.y);
}

// ---- the read-out with 8-bit sub-digits (the shipped form; H2_READOUT_NIBBLES=1 keeps the one above for A/B) ----------------
// |d_w| = e_0 + 256 e_1 with e_0 in [-127, 128], e_1 in [0, 128]: 2 x 128 lists instead of 4 x 8, and 2^J * 32 terms per output
// instead of 2^J * 64.  So that the 256 bucket sums per output neither travel through memory nor cost a lane each, a lane owns
// SIXTEEN consecutive magnitudes of one position (workgroup row y = position * 8 + g: magnitudes 16 g + 1 .. 16 g + 16), walks
// their lists from the largest down and keeps the two running sums of the bucket method in registers:
//     run += B_v;  tot += run      =>      tot = sum_r r * B_{16 g + r},   run = sum_r B_{16 g + r}
// (control flow and list reads are uniform across a wave, the gathers coalesced, as above).  ipa_readout_combine then forms, per
// output and position, sum_g tot_g + 16 * sum_g g * run_g (a second running sum over the 8 groups), ipa_readout_finish
// P_0 + 256 P_1 and the affine result.  Per output at J = 6: ~1800 mixed additions + 512 full ones against ~3800 + 64.
template <int FB>
__global__ void __launch_bounds__(256, 2) ipa_readout_groups(const u32 *__restrict__ table, const u32 *__restrict__ list,
                                                             const u32 *__restrict__ list_start, u32 nJ, u32 *__restrict__ sums) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (i >= nJ) return;
    const u32 first = (y >> 3) * 128 + (y & 7) * 16;          // the list of magnitude 16 g + 1 at this position
    xyzz9<FB> run = xyzz9_identity<FB>(), tot = xyzz9_identity<FB>();
    for (int r = 15; r >= 0; --r) {
        const u32 lo = list_start[first + r], hi = list_start[first + r + 1];
        xyzz9<FB> acc = xyzz9_identity<FB>();
        if (lo < hi) {
            u32 e0 = list[lo], e1 = list[min(lo + 1, hi - 1)];
            affine<FB> nxt = aff_load<FB>(table + 16 * ((size_t)(e0 & 0x7FFFFFFFu) + i));
            for (u32 t = lo; t < hi; ++t) {
                const affine<FB> p = nxt;
                const u32 neg = e0 >> 31;
                const u32 e2 = list[min(t + 2, hi - 1)];
                nxt = aff_load<FB>(table + 16 * ((size_t)(e1 & 0x7FFFFFFFu) + i));
                e0 = e1;
                e1 = e2;
                if (!aff_is_identity(p)) {
                    aff9<FB> q = aff9_unpack<FB>(p);
                    if (neg) q.y = fe9_sub(fe9_zero(), q.y);
                    xyzz9_madd<FB>(acc, q);
                }
            }
        }
        xyzz9_add<FB>(run, acc);
        xyzz9_add<FB>(tot, run);
    }
    xyzz_store<FB>(sums + 32 * ((size_t)(2 * y) * nJ + i), xyzz9_is_identity(tot) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(tot));
    xyzz_store<FB>(sums + 32 * ((size_t)(2 * y + 1) * nJ + i), xyzz9_is_identity(run) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(run));
}
// quad (i, position): sums[32 + position][i] <- sum_g tot_g + 16 * sum_g g * run_g.  27 dependent point operations per output and position,
// 2 nJ chains: the chip is nearly idle in them, so each runs on a QUAD of lanes (curve_wide.cuh; round 5: 0.22 -> ~0.08 ms at 2^14 outputs)
template <int FB>
__global__ void __launch_bounds__(256) ipa_readout_combine(u32 *__restrict__ sums, u32 nJ) {
    H2_LATENCY_STAGE();
    const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (t >= 2 * nJ) return;
    const u32 pos = t / nJ, i = t % nJ;
    xyzz<FB> P = xyzz_identity<FB>(), rr = xyzz_identity<FB>(), tt = xyzz_identity<FB>();
    for (int g = 7; g >= 0; --g) {
        const u32 y = pos * 8 + g;
        xyzz_add_wide<FB>(P, xyzz_load<FB>(sums + 32 * ((size_t)(2 * y) * nJ + i)));
        if (g) {
            xyzz_add_wide<FB>(rr, xyzz_load<FB>(sums + 32 * ((size_t)(2 * y + 1) * nJ + i)));
            xyzz_add_wide<FB>(tt, rr);
        }
    }
    for (int d = 0; d < 4; ++d) tt = xyzz_dbl_wide<FB>(tt);
    xyzz_add_wide<FB>(P, tt);
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz_store<FB>(sums + 32 * ((size_t)(32 + pos) * nJ + i), P);
}
template <int FB>
__global__ void __launch_bounds__(256) ipa_readout_finish(const u32 *__restrict__ sums, u32 nJ, u32 *__restrict__ out_xy) {
    H2_LATENCY_STAGE();
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (i >= nJ) return;
    xyzz<FB> acc = xyzz_load<FB>(sums + 32 * ((size_t)33 * nJ + i));
    for (int d = 0; d < 8; ++d) acc = xyzz_dbl_wide<FB>(acc);
    xyzz_add_wide<FB>(acc, xyzz_load<FB>(sums + 32 * ((size_t)32 * nJ + i)));
    if ((threadIdx.x & (kGroup - 1)) != 0) return;
    const affine<FB> a = xyzz_to_affine<FB>(acc);
    fe_store(out_xy + 16 * (size_t)i, a.x);
    fe_store(out_xy + 16 * (size_t)i + 8, a.y);
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
static bool timeline_on() {
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

// ---- host orchestration ----------------------------------------------------------------------------
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
    }
    bool attr_set = false, attr2_set = false, attr_bins_set = false;
    hipStream_t copy_stream = nullptr;      // h2_msm: the bases cross PCIe on this one while the sort runs (null-stream context only)
    hipEvent_t copy_done = nullptr;
    // the slice split of a large generic multiexp (msm_launch): the upper slices' fold and Horner chain run on `side` beside the lower
    // slices' accumulate; fork / conv / acc_a / join order the two streams
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_conv = nullptr, ev_acc_a = nullptr, ev_join = nullptr;
    u32 lanes[2][3] = {{0, 0, 0}, {0, 0, 0}};  // resident lanes of msm_accumulate<FP / FQ, plain / GLV> on this device
};

// One workspace per (device, stream): calls enqueued on different streams never share scratch.
static std::mutex g_ctx_mu;
static std::map<std::pair<int, hipStream_t>, std::unique_ptr<MsmContext>> g_ctxs;
static MsmContext &msm_ctx(hipStream_t st = nullptr) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &slot = g_ctxs[std::make_pair(dev, st)];
    if (!slot) slot.reset(new MsmContext());
    return *slot;
}
// h2_trim: the per-(device, stream) scratch of this device goes back to the allocator (the device is idle by then)
void msm_release_host_pipe();
void msm_release_host_msm_pipe();
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
static constexpr int H2_ERR_BATCH_SHAPE = -1000;     // internal: never leaves this file

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
    static const bool glv_on_fe9 = [] { const char *e = getenv("H2_GENERIC_FE9"); return !(e && atoi(e) == 0); }();
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
        static const int acc_waves = [] { const char *e = getenv("H2_ACC_WAVES"); int v = e ? atoi(e) : 0; return v >= 1 && v <= 4 ? v : H2_ACC9_WAVES; }();
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
    static const u32 oversub = [] { const char *e = getenv("H2_ACC_OVERSUB"); int v = e ? atoi(e) : 0; return (u32)(v >= 1 && v <= 8 ? v : 1); }();
    const u32 usable = std::max(256u, (u32)(lanes * fraction) / 256u * 256u) * (a.table ? oversub : 1u);
    // entries per lane of the accumulate: 16 for full-size columns; small commits are chains of latency-bound kernels and run
    // shorter with more, shorter lanes (one registered commit at 2^11 .. 2^15 points: 3-7 % faster at 8; H2_MSM_DIV: sweeps only)
    static const u32 env_div = [] { const char *e = getenv("H2_MSM_DIV"); int v = e ? atoi(e) : 0; return (u32)(v >= 1 && v <= 64 ? v : 0); }();
    const u32 lane_div = env_div ? env_div : (all_items < ((size_t)1 << 20) ? 8u : 16u);
    // Column-batched commits are JOINED (ColStride::joined) unless H2_BATCH_JOIN=0: the K sorted lists form one, which ONE launch of
    // msm_accumulate cuts into equal ranges -- the chip is tiled exactly as by a single commit (a launch per column leaves its last
    // round of workgroups ragged, and K of them next to each other share CUs unevenly), and the finisher meets T range heads per
    // batch instead of per column.
    static const bool join_env = [] { const char *e = getenv("H2_BATCH_JOIN"); return !(e && e[0] == '0'); }();
    const bool joined = K > 1 && join_env;
    u32 T = (u32)std::min<size_t>(usable, std::max<size_t>(256, ((joined ? K : 1) * all_items / lane_div + 255) / 256 * 256));
    size_t head_slots = joined ? (size_t)T : (size_t)T * K;            // range heads parked in cx.seg9, in front of the K x tb bucket slots
    const u32 max_heavy = kMaxHeavy;
    // two-pass sort (registered path): always for windows beyond 16 bits, else for large bucket counts
    Sort2 S2;
    memset(&S2, 0, sizeof S2);
    S2.pair_shift = -1;
    static const u32 run_lanes_env = [] { const char *e = getenv("H2_S1_RUN_LANES"); int v = e ? atoi(e) : 0; return (u32)(v == 8 || v == 16 || v == 32 || v == 64 ? v : 16); }();
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
        static const int force_old = [] { const char *e = getenv("H2_MSM_SORT"); return e && atoi(e) == 1 ? 1 : 0; }();
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
        static const int force_old = [] { const char *e = getenv("H2_MSM_SORT"); return e && atoi(e) == 1 ? 1 : 0; }();
        int lb = 0, kb = 0;
        while (((u64)(m - 1) >> lb) != 0) ++lb;
        while (((u64)(tb - 1) >> kb) != 0) ++kb;
        // bins of ~16 K entries (kb - 11 bucket bits per bin: 1152 bins for 9 slices of 2^15 buckets), so that pass 2 is the
        // one-launch form with a bin per workgroup in LDS; H2_GLV_BIN_BITS: sweeps only (9 = the chunked pass 2 of round 2)
        // Up to 2^19 scalars only: the carry slice of the split (the window above the top of a 128-bit half) puts ~n / 2 entries
        // into ONE bucket, and a bin that large was scattered by a single workgroup (2^19: sort 0.28 -> 0.16 ms; 2^20: 0.30 -> 0.61).
        // With the oversized-bin kernels (msm_s2_big_*) that bin is chunked over 64 workgroups: 2^20 takes the one-launch form with
        // 10 bits (1.83 -> 1.71 ms on one box; 11 bits 1.80, 12 bits 1.83); from 2^21 the forms are equal within 1 %.
        static const int glv_bin_bits = [] { const char *e = getenv("H2_GLV_BIN_BITS"); int v = e ? atoi(e) : 0; return v >= 8 && v <= 12 ? v : 0; }();
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
    static const bool fold9_on = [] { const char *e = getenv("H2_FOLD9"); return !(e && atoi(e) == 0); }();     // A/B switch
    // the fold on the carry-free layer (fold9_* kernels: registered tables from 16-bit windows, paired commits, and the window
    // slices of a large generic multiexp); a range of a chunked commit hands finished buckets on in the reference's form
    // (add_into), so it keeps the 8 x 32 finisher
    static const u32 fold9_min_nb = [] { const char *e = getenv("H2_FOLD9_MIN_NB"); int v = e ? atoi(e) : 0; return (u32)(v >= 64 ? v : 128); }();
    const bool fold9 = fold9_on && sh.NB >= fold9_min_nb && sh.slices <= 16 && m9 && !a.add_into && !fold_only;      // (16: arrival counters of fold9_planes)
    if (K > 1 && !fold9) return H2_ERR_BATCH_SHAPE;
    if (a.slice_sums_only && !(fold9 && glv)) return H2_ERR_BATCH_SHAPE;
    // Slice split (round 5; generic multiexps from 2^19 points): the sorted list is ordered by (slice, bucket), so the upper slices
    // [split_k, slices) and the lower ones [0, split_k) are two contiguous halves of it.  They are accumulated one after the other on
    // `st`; as soon as the UPPER group is in its buckets its fold and its Horner chain -- (slices - 1) c ~ 128 dependent doublings, 0.25 ms
    // on one quad of lanes, which used to follow the whole accumulate -- run on a side stream beside the lower group's accumulate and
    // fold.  What is left behind the accumulate: the lower group's fold, (split_k - 1) c doublings and one addition.  The bases'
    // conversion to M9 form runs on the side stream beside the sort.  H2_GENERIC_SPLIT=0: off (A/B); = k: force the lower group's size.
    static const int split_env = [] { const char *e = getenv("H2_GENERIC_SPLIT"); return e ? atoi(e) : -1; }();
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
        static const bool bins_on = [] { const char *e = getenv("H2_S2_BINS"); return !(e && atoi(e) == 0); }();
        const size_t nbk = (size_t)1 << S2.lowb;
        // LDS stage: the average bin + 25 % (two workgroups per CU where that fits: 2^20 scalars at 17 bits, 15 K-entry bins), at
        // most what one workgroup can have; a bin beyond its stage takes the direct-scatter branch.  H2_S2_CAP: sweeps only.
        const size_t cap_max = nbk * 8 + 64 < kLdsCap ? (kLdsCap - nbk * 8) / 4 : 0;
        static const size_t cap_env = [] { const char *e = getenv("H2_S2_CAP"); return e ? (size_t)atol(e) : (size_t)0; }();
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
        static const u32 s1_threads = [] { const char *e = getenv("H2_S1_THREADS"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : 512); }();
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
            static const u32 big_threads = [] { const char *e = getenv("H2_S2_BIG_THREADS"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : 256); }();
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

static int msm_dispatch(MsmContext &cx, int curve, const MsmArgs &a, hipStream_t st) {
    if (curve == H2_PALLAS) return msm_launch<FP, FQ>(cx, a, st);
    return msm_launch<FQ, FP>(cx, a, st);
}

static void to_mont_async(int curve, u32 *d, size_t field_elems, hipStream_t st) {
    if (!field_elems) return;
    dim3 grid((unsigned)((field_elems + 255) / 256)), block(256);
    if (curve == H2_PALLAS) hipLaunchKernelGGL((k_to_mont<FP>), grid, block, 0, st, d, field_elems);
    else hipLaunchKernelGGL((k_to_mont<FQ>), grid, block, 0, st, d, field_elems);
}

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
static std::mutex g_bases_mu;
static std::map<h2_bases_t, std::shared_ptr<Bases>> g_bases;
static h2_bases_t g_next_handle = 1;

// endomorphism: may the handle be an ENDOMORPHISM table (Bases::glv)?  Such a table belongs to the opening argument's round loop and never leaves the
// library; only the sub-digit paired commit, its refill and the bookkeeping entry points read it -- to everything else it is not a handle.
static std::shared_ptr<Bases> find_bases(h2_bases_t h, bool endomorphism = false) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(h);
    if (it == g_bases.end() || (it->second->glv && !endomorphism)) return nullptr;
    return it->second;
}

static bool bad_common(int curve, int form, int out_kind) {
    return (curve != H2_PALLAS && curve != H2_VESTA) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) ||
           (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE);
}

// fills rows 1..W-1 of the table for columns [first, first + count) from row 0
// keep_tmp: the staging stays with the handle (a refilled table pays hipMalloc / hipFree -- ~0.3 ms, and a device synchronisation each -- once)
static int table_fill(Bases &b, u32 first, u32 count, hipStream_t st, bool keep_tmp = false) {
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
static int set_blind_base_host(Bases &b, const void *host_w_xy, int form) {
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
static int override_blind_base_device(Bases &b, const void *d_w_xy, int form, hipStream_t st) {
    std::lock_guard<std::mutex> lk(b.mu);
    launch_blind_install(b, d_w_xy, form, st);
    H2_HIP(hipGetLastError());
    b.blind_set = true;
    b.blind_host_known = false;
    return H2_OK;
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

// ---- h2_msm from host slices, large: the multiexp is cut into point RANGES that start as their bases land ------------------------
// 96 bytes per point cross PCIe (2^20 points: ~1.8 ms) and the multiexp behind them is ~1.5 ms of device time; run one after the
// other (round 4: 3.45 ms with only the sort under the upload) the accumulate waits for the LAST base.  Here the scalars cross first,
// every range's sort is enqueued at once (it reads scalars only), then the bases cross range by range and range q's conversion +
// accumulate run behind the event of ITS bases: sum_i k_i P_i = sum_q (sum_{i in range q} k_i P_i).  Three streams: `copy` (the
// bases), `heavy` (sorts, conversions, accumulates: the full-chip kernels, in range order) and `light` (each range's fold down to its
// window-slice sums -- short latency-bound launches that must not sit in front of the next accumulate -- and, at the end, ONE Horner
// step over the slice sums of all ranges, msm_combine_ranges: the 128-doubling chain is paid once, not per range).
// The pageable host-to-device copies hold the calling thread, and every launch enqueued between two of them is PCIe idle time (a
// helper thread does not help: launches and a pageable copy contend inside the runtime, bench/ubench_h2d.hip), so the launch
// sequences are captured ONCE per shape as hipGraphs -- one for the sorts, one per range for its accumulate and for its fold, one for
// the final step -- and replayed with one call each (profiles/r05_h2_msm_host_ranges.txt).  A graph bakes in its kernels' pointer
// arguments: it is replayed only while every buffer it names is where it was (DevBuf epoch), and never while the event profiler or the
// debug timeline is on.  What stays exposed behind the upload: the LAST range's accumulate, its fold and the Horner step (~0.75 ms).
namespace {
struct HostMsmPipe {
    std::mutex mu;
    hipStream_t copy = nullptr, heavy = nullptr, light = nullptr;
    hipEvent_t scalars_in = nullptr, folds_done = nullptr;
    std::vector<hipEvent_t> landed, acc_done;
    std::vector<std::unique_ptr<MsmContext>> ctx;       // one workspace per range (they share the `heavy` stream, not scratch)
    // captured launch sequences of the last shape seen
    struct Shape {
        int curve = -1, form = -1, out_kind = -1, c = 0;
        size_t n = 0;
        unsigned Q = 0;
        const void *s = nullptr, *b = nullptr, *o = nullptr;
        unsigned long epoch = 0;
        bool operator==(const Shape &x) const {
            return curve == x.curve && form == x.form && out_kind == x.out_kind && c == x.c && n == x.n && Q == x.Q && s == x.s && b == x.b && o == x.o && epoch == x.epoch;
        }
    } shape;
    int warm = 0;                                       // calls seen with `shape`: the first runs plain launches (allocations, attributes), the second captures
    hipGraphExec_t g_sort = nullptr, g_final = nullptr;
    std::vector<hipGraphExec_t> g_acc, g_fold;
    void drop_graphs() {
        if (g_sort) (void)hipGraphExecDestroy(g_sort);
        if (g_final) (void)hipGraphExecDestroy(g_final);
        for (auto g : g_acc) if (g) (void)hipGraphExecDestroy(g);
        for (auto g : g_fold) if (g) (void)hipGraphExecDestroy(g);
        g_sort = g_final = nullptr;
        g_acc.clear();
        g_fold.clear();
    }
    int ensure(unsigned q) {
        if (!copy) {
            H2_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
            H2_HIP(hipStreamCreateWithFlags(&heavy, hipStreamNonBlocking));
            H2_HIP(hipStreamCreateWithFlags(&light, hipStreamNonBlocking));
            H2_HIP(hipEventCreateWithFlags(&scalars_in, hipEventDisableTiming));
            H2_HIP(hipEventCreateWithFlags(&folds_done, hipEventDisableTiming));
        }
        while (landed.size() < q) {
            hipEvent_t a, b;
            H2_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            H2_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            landed.push_back(a);
            acc_done.push_back(b);
            ctx.emplace_back(new MsmContext());
        }
        return H2_OK;
    }
};
HostMsmPipe g_host_msm[16];      // per device
}  // namespace
// ranges of a host-pointer multiexp of n points: three from 2^19 points on (2^20: 2 / 3 / 4 / 8 ranges = 3.08-3.19 / 3.07-3.13 /
// 3.11-3.26 / 3.63 ms against 3.44-3.46 in one piece: finer ranges lose more to the per-copy cost of pageable memory and to narrower
// windows than their shorter tail wins -- profiles/r05_h2_msm_host_ranges.txt; H2_MSM_HOST_CHUNKS: sweeps, 1 = the round-4 path)
static unsigned host_msm_chunks(size_t n) {
    static const int env = [] { const char *e = getenv("H2_MSM_HOST_CHUNKS"); return e ? atoi(e) : 0; }();
    if (env >= 1) return (unsigned)std::min<size_t>((size_t)std::min(env, 16), std::max<size_t>(1, n >> 14));
    if (n < ((size_t)1 << 19)) return 1;
    return 3;
}
// runs `body` (launches on `st`) either directly or into a new executable graph
template <class Body> static int capture_graph(hipStream_t st, hipGraphExec_t *exec, Body body) {
    H2_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    const int rc = body();
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != H2_OK || e != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        if (rc != H2_OK) return rc;
        set_last_hip_error(e, __FILE__, __LINE__);
        return H2_ERR_HIP;
    }
    const hipError_t ei = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) { set_last_hip_error(ei, __FILE__, __LINE__); return H2_ERR_HIP; }
    return H2_OK;
}
// H2_ERR_BATCH_SHAPE: the ranges do not take the slice-sum form (nothing was enqueued): the caller falls back to the one-piece path
static int msm_host_chunked(MsmContext &cx, int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, int form, int out_kind,
                            unsigned Q) {
    int dev = 0;
    H2_HIP(hipGetDevice(&dev));
    HostMsmPipe &hp = g_host_msm[dev & 15];
    std::lock_guard<std::mutex> lk(hp.mu);
    int rc = hp.ensure(Q);
    if (rc != H2_OK) return rc;
    auto range = [&](unsigned q, size_t &lo, size_t &hi) { lo = n * q / Q; hi = n * (q + 1) / Q; };
    const int c = choose_c((n + Q - 1) / Q, false);            // ONE window width for every range: their slice sums add up
    auto args_of = [&](unsigned q, int phase) {
        size_t lo, hi;
        range(q, lo, hi);
        MsmArgs a{(const char *)cx.stage_s.ptr + 32 * lo, nullptr, (const char *)cx.stage_b.ptr + 64 * lo, nullptr, hi - lo, false,
                  c, 0, 0xFFFFFFFFu, form, H2_OUT_JACOBIAN, nullptr};
        a.phase = phase;
        a.slice_sums_only = true;
        return a;
    };
    auto stage = [&](unsigned q, int phase, hipStream_t st) {
        std::lock_guard<std::mutex> lq(hp.ctx[q]->mu);
        return msm_dispatch(*hp.ctx[q], curve, args_of(q, phase), st);
    };
    int slices = 0;
    {
        size_t lo, hi;
        range(0, lo, hi);
        slices = (int)make_shape(2 * (hi - lo), c, false, true).slices;
    }
    auto final_step = [&]() {
        RangeSums rs;
        memset(&rs, 0, sizeof rs);
        for (unsigned q = 0; q < Q; ++q) rs.p[q] = hp.ctx[q]->ssums.as<u32>();
        if (curve == H2_PALLAS) hipLaunchKernelGGL((msm_combine_ranges<FP>), dim3(1), dim3(64), 0, hp.light, rs, (int)Q, slices, c, cx.out.as<u32>(), out_kind, form == H2_FORM_MONTGOMERY);
        else hipLaunchKernelGGL((msm_combine_ranges<FQ>), dim3(1), dim3(64), 0, hp.light, rs, (int)Q, slices, c, cx.out.as<u32>(), out_kind, form == H2_FORM_MONTGOMERY);
        H2_HIP(hipGetLastError());
        return (int)H2_OK;
    };
    // the shape probe: does a range take the slice-sum form?  (phase 1 on an EMPTY capture would be clumsy: ask the planner directly --
    // the generic path folds on the carry-free layer from 128 buckets per slice on, with at most 16 slices)
    {
        size_t lo, hi;
        range(0, lo, hi);
        const MsmShape sh = make_shape(2 * (hi - lo), c, false, true);
        if (!glv_applies(hi - lo) || sh.NB < 128 || sh.slices > 16 || Q > 16) return H2_ERR_BATCH_SHAPE;
    }
    HostMsmPipe::Shape want;
    want.curve = curve; want.form = form; want.out_kind = out_kind; want.c = c; want.n = n; want.Q = Q;
    want.s = cx.stage_s.ptr; want.b = cx.stage_b.ptr; want.o = cx.out.ptr;
    want.epoch = devbuf_epoch();
    static const bool graphs_on = [] { const char *e = getenv("H2_MSM_HOST_GRAPHS"); return !(e && atoi(e) == 0); }();
    if (!(want == hp.shape)) {
        hp.drop_graphs();
        hp.shape = want;
        hp.warm = 0;
    }
    const bool may_graph = graphs_on && !prof_enabled() && !timeline_on();
    if (may_graph && hp.warm >= 1 && !hp.g_sort) {
        // second call with this shape: every workspace exists, every attribute is set -- capture the launch sequences (nothing executes)
        hp.g_acc.assign(Q, nullptr);
        hp.g_fold.assign(Q, nullptr);
        rc = capture_graph(hp.heavy, &hp.g_sort, [&] { int r = H2_OK; for (unsigned q = 0; q < Q && r == H2_OK; ++q) r = stage(q, 1, hp.heavy); return r; });
        for (unsigned q = 0; q < Q && rc == H2_OK; ++q) {
            rc = capture_graph(hp.heavy, &hp.g_acc[q], [&] { return stage(q, 3, hp.heavy); });
            if (rc == H2_OK) rc = capture_graph(hp.light, &hp.g_fold[q], [&] { return stage(q, 4, hp.light); });
        }
        if (rc == H2_OK) rc = capture_graph(hp.light, &hp.g_final, final_step);
        if (rc != H2_OK || devbuf_epoch() != want.epoch) {     // (a capture that had to allocate is not replayable: stay with plain launches)
            hp.drop_graphs();
            hp.shape.epoch = devbuf_epoch();
            if (rc != H2_OK) return rc;
        }
    }
    const bool replay = may_graph && hp.g_sort != nullptr;
    // Who enqueues: with the captured sequences a call is ~4 + 3 Q runtime calls; a helper thread CAN make them while this thread
    // goes from one pageable copy straight into the next (each event is recorded here, right behind its copy; the helper picks it up
    // through an atomic counter).
    // Measured (profiles/r05_h2_msm_host_ranges.txt, 2^20): no gain -- 3.08-3.27 ms with the helper against 3.07-3.23 without: what the
    // copies lose to the helper's calls is what they idled before.  Off unless H2_MSM_HOST_THREAD=1.
    static const bool thread_on = [] { const char *e = getenv("H2_MSM_HOST_THREAD"); return e && atoi(e) == 1; }();
    const bool helper = replay && thread_on;
    std::atomic<int> landed_n{-1};          // -1: nothing yet; 0: the scalars' event is recorded; q + 1: range q's
    std::atomic<int> abort_flag{0};
    int helper_rc = H2_OK;
    auto enqueue_sorts = [&]() -> int {
        hipError_t e = hipStreamWaitEvent(hp.heavy, hp.scalars_in, 0);
        if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
        if (replay) {
            if ((e = hipGraphLaunch(hp.g_sort, hp.heavy)) != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
            return H2_OK;
        }
        int r = H2_OK;
        for (unsigned q = 0; q < Q && r == H2_OK; ++q) r = stage(q, 1, hp.heavy);       // every range's sort: it needs the scalars only
        return r;
    };
    auto enqueue_range = [&](unsigned q) -> int {      // range q's accumulate on `heavy` behind its bases, its fold on `light`
        int r = H2_OK;
        hipError_t e = hipStreamWaitEvent(hp.heavy, hp.landed[q], 0);
        if (e == hipSuccess) {
            if (replay) e = hipGraphLaunch(hp.g_acc[q], hp.heavy);
            else r = stage(q, 3, hp.heavy);
        }
        if (r == H2_OK && e == hipSuccess) e = hipEventRecord(hp.acc_done[q], hp.heavy);
        if (r == H2_OK && e == hipSuccess) e = hipStreamWaitEvent(hp.light, hp.acc_done[q], 0);
        if (r == H2_OK && e == hipSuccess) {
            if (replay) e = hipGraphLaunch(hp.g_fold[q], hp.light);
            else r = stage(q, 4, hp.light);
        }
        if (r == H2_OK && e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); r = H2_ERR_HIP; }
        return r;
    };
    auto enqueue_final = [&]() -> int {
        hipError_t e = hipSuccess;
        int r = H2_OK;
        if (replay) {
            if ((e = hipGraphLaunch(hp.g_final, hp.light)) != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
        } else if ((r = final_step()) != H2_OK) {
            return r;
        }
        if ((e = hipEventRecord(hp.folds_done, hp.light)) != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
        return H2_OK;
    };
    std::thread worker;
    if (helper)
        worker = std::thread([&] {
            (void)hipSetDevice(dev);
            auto wait_for = [&](int v) {
                while (landed_n.load(std::memory_order_acquire) < v && !abort_flag.load(std::memory_order_acquire)) std::this_thread::yield();
                return !abort_flag.load(std::memory_order_acquire);
            };
            if (!wait_for(0)) return;
            int r = enqueue_sorts();
            for (unsigned q = 0; q < Q && r == H2_OK; ++q) {
                if (!wait_for((int)q + 1)) return;
                r = enqueue_range(q);
            }
            if (r == H2_OK) r = enqueue_final();
            helper_rc = r;
        });
    hipError_t e = hipStreamSynchronize(0);                    // the staging buffers may still be read by an earlier call's kernels
    if (e == hipSuccess) e = hipMemcpyAsync(cx.stage_s.ptr, scalars, n * 32, hipMemcpyHostToDevice, 0);
    if (e == hipSuccess) e = hipEventRecord(hp.scalars_in, 0);
    if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); rc = H2_ERR_HIP; }
    if (rc == H2_OK) {
        if (helper) landed_n.store(0, std::memory_order_release);
        else rc = enqueue_sorts();
    }
    // range q's bases: the pageable copy holds this thread for its length
    for (unsigned q = 0; q < Q && rc == H2_OK; ++q) {
        size_t lo, hi;
        range(q, lo, hi);
        e = hipMemcpyAsync((char *)cx.stage_b.ptr + 64 * lo, (const char *)bases_xy + 64 * lo, 64 * (hi - lo), hipMemcpyHostToDevice, hp.copy);
        if (e == hipSuccess && form == H2_FORM_CANONICAL) to_mont_async(curve, (u32 *)((char *)cx.stage_b.ptr + 64 * lo), (hi - lo) * 2, hp.copy);
        if (e == hipSuccess) e = hipEventRecord(hp.landed[q], hp.copy);
        if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); rc = H2_ERR_HIP; break; }
        if (helper) landed_n.store((int)q + 1, std::memory_order_release);
        else rc = enqueue_range(q);
    }
    if (helper) {
        if (rc != H2_OK) abort_flag.store(1, std::memory_order_release);
        worker.join();
        if (rc == H2_OK) rc = helper_rc;
    } else if (rc == H2_OK) {
        rc = enqueue_final();
    }
    if (rc == H2_OK && (e = hipStreamWaitEvent(0, hp.folds_done, 0)) != hipSuccess) {
        set_last_hip_error(e, __FILE__, __LINE__);
        rc = H2_ERR_HIP;
    }
    if (rc != H2_OK) {            // drain everything that was enqueued before reporting
        (void)hipStreamSynchronize(hp.heavy);
        (void)hipStreamSynchronize(hp.light);
        (void)hipStreamSynchronize(hp.copy);
        (void)hipStreamSynchronize(0);
        return rc;
    }
    hp.warm++;
    return H2_OK;
}
namespace h2 {
void msm_release_host_msm_pipe() {        // h2_trim: the ranges' workspaces and the graphs that name them
    int dev = 0;
    (void)hipGetDevice(&dev);
    HostMsmPipe &hp = g_host_msm[dev & 15];
    std::lock_guard<std::mutex> lk(hp.mu);
    hp.drop_graphs();
    hp.shape = HostMsmPipe::Shape();
    hp.warm = 0;
    for (auto &c : hp.ctx) {
        std::lock_guard<std::mutex> lc(c->mu);
        c->release_all();
    }
}
}  // namespace h2

extern "C" int h2_msm(int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, int form, int out_kind,
                      uint64_t *out) {
    if (bad_common(curve, form, out_kind) || !out || (n && (!scalars || !bases_xy)) || n > 0x7FFFFFF0u) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    MsmContext &cx = msm_ctx();
    const size_t out_bytes = out_kind == H2_OUT_AFFINE ? 64 : 96;
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.stage_s.reserve(n * 32 + 32)) != H2_OK) return rc;
    if ((rc = cx.stage_b.reserve(n * 64 + 64)) != H2_OK) return rc;
    if ((rc = cx.out.reserve(128)) != H2_OK) return rc;
    if (const unsigned Q = host_msm_chunks(n); Q > 1) {
        rc = msm_host_chunked(cx, curve, scalars, bases_xy, n, form, out_kind, Q);
        if (rc == H2_OK) {
            H2_HIP(hipMemcpyAsync(out, cx.out.ptr, out_bytes, hipMemcpyDeviceToHost, 0));
            H2_HIP(hipStreamSynchronize(0));
            return H2_OK;
        }
        if (rc != H2_ERR_BATCH_SHAPE) return rc;              // (a shape the range pipeline does not take: one piece, below)
    }
    MsmArgs a{cx.stage_s.ptr, nullptr, cx.stage_b.ptr, nullptr, n, false, choose_c(n ? n : 1, false), 0, 0xFFFFFFFFu, form,
              out_kind, cx.out.ptr};
    // Large multiexps: the scalars cross first (a third of the bytes), the sort -- which reads nothing else -- is enqueued, and only
    // then does the host enter the copy of the bases, on a second stream: the sort runs while the bases are on the bus (2^20 points:
    // 0.25 ms of a 3.6 ms call).  H2_MSM_HOST_OVERLAP=0: copy, copy, compute (A/B).
    static const bool overlap_on = [] { const char *e = getenv("H2_MSM_HOST_OVERLAP"); return !(e && atoi(e) == 0); }();
    if (n >= ((size_t)1 << 16) && overlap_on) {
        if (!cx.copy_stream) {
            H2_HIP(hipStreamCreateWithFlags(&cx.copy_stream, hipStreamNonBlocking));
            H2_HIP(hipEventCreateWithFlags(&cx.copy_done, hipEventDisableTiming));
        }
        H2_HIP(hipStreamSynchronize(0));                       // the staging buffers may still be read by an earlier call's kernels
        H2_HIP(hipMemcpyAsync(cx.stage_s.ptr, scalars, n * 32, hipMemcpyHostToDevice, 0));
        a.phase = 1;
        if ((rc = msm_dispatch(cx, curve, a, 0)) != H2_OK) { (void)hipStreamSynchronize(0); return rc; }
        hipError_t e = hipMemcpyAsync(cx.stage_b.ptr, bases_xy, n * 64, hipMemcpyHostToDevice, cx.copy_stream);
        if (e == hipSuccess && form == H2_FORM_CANONICAL) to_mont_async(curve, cx.stage_b.as<u32>(), n * 2, cx.copy_stream);
        if (e == hipSuccess) e = hipEventRecord(cx.copy_done, cx.copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(0, cx.copy_done, 0);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(cx.copy_stream);
            (void)hipStreamSynchronize(0);
            set_last_hip_error(e, __FILE__, __LINE__);
            return H2_ERR_HIP;
        }
        a.phase = 2;
        rc = msm_dispatch(cx, curve, a, 0);
        if (rc != H2_OK) { (void)hipStreamSynchronize(0); return rc; }
    } else {
        if (n) {
            H2_HIP(hipMemcpyAsync(cx.stage_s.ptr, scalars, n * 32, hipMemcpyHostToDevice, 0));
            H2_HIP(hipMemcpyAsync(cx.stage_b.ptr, bases_xy, n * 64, hipMemcpyHostToDevice, 0));
            if (form == H2_FORM_CANONICAL) to_mont_async(curve, cx.stage_b.as<u32>(), n * 2, 0);
        }
        if ((rc = msm_dispatch(cx, curve, a, 0)) != H2_OK) return rc;
    }
    H2_HIP(hipMemcpyAsync(out, cx.out.ptr, out_bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

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
    if (const char *e = getenv("H2_COLUMN_C")) {          // sweeps only (bench.py, bench/tools): the width column tables are built with
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
    static const long v = [] { const char *e = getenv("H2_PAIR_SUBDIGITS"); return e ? atol(e) : (long)((1 << 16) + 4); }();
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

extern "C" int h2_ipa_collapsed_generators_device(h2_bases_t basis, unsigned k, unsigned rounds, const uint64_t *challenges, int form,
                                                  void *d_out_xy, void *stream) {
    auto b = find_bases(basis);
    if (!b) return H2_ERR_HANDLE;
    if ((form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !challenges || !d_out_xy || k < 1 || k > 26 || rounds < 1 ||
        rounds > k || rounds > 12 || b->n < ((size_t)1 << k) || b->c != 16 || b->W != 16)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int sf = b->curve == H2_PALLAS ? H2_FQ : H2_FP;
    const u32 J = rounds, nJ = 1u << (k - J);
    u64 um[12 * 4];
    for (u32 r = 0; r < J; ++r) host_to_mont(sf, um + 4 * r, challenges + 4 * r, form);
    static const bool nibbles = [] { const char *e = getenv("H2_READOUT_NIBBLES"); return e && e[0] == '1'; }();
    const int nlists = nibbles ? 32 : 256;
    std::vector<std::vector<u32>> lists(nlists);
    for (u32 h = 0; h < (1u << J); ++h) {
        u64 s[4], canon[4];
        memcpy(s, kHostField[sf].one, 32);
        for (u32 r = 0; r < J; ++r)
            if ((h >> (J - 1 - r)) & 1) host_mul(sf, s, s, um + 4 * r);          // the products of ipa_s_table
        host_from_mont(sf, canon, s);
        u32 carry = 0;
        for (u32 w = 0; w < 16; ++w) {
            u32 raw = (u32)((canon[w >> 2] >> (16 * (w & 3))) & 0xFFFFu) + carry;   // signed 16-bit digits, as msm_recode cuts them
            const bool neg = raw > 0x8000u;
            carry = neg ? 1 : 0;
            u32 mag = neg ? 0x10000u - raw : raw;                                 // |d| <= 2^15
            const u32 off = w * b->stride + h * nJ;
            if (!nibbles) {                                                       // |d| = e_0 + 256 e_1, e_0 in [-127, 128], e_1 in [0, 128]
                u32 e0 = mag & 255u, c8 = 0;
                bool e0neg = false;
                if (e0 > 128) {
                    e0 = 256 - e0;
                    e0neg = true;
                    c8 = 1;
                }
                const u32 e1 = (mag >> 8) + c8;                                   // <= 128: mag <= 2^15, and mag = 2^15 has e_0 = 0
                if (e0) lists[e0 - 1].push_back(off | ((neg != e0neg) ? 0x80000000u : 0u));
                if (e1) lists[128 + e1 - 1].push_back(off | (neg ? 0x80000000u : 0u));
                continue;
            }
            u32 c4 = 0;
            for (u32 v = 0; v < 4; ++v) {                                         // |d| = sum_v 16^v e_v, e_v in [-7, 8]
                u32 e = ((mag >> (4 * v)) & 15u) + c4;
                bool eneg = false;
                c4 = 0;
                if (e > 8) {
                    e = 16 - e;
                    eneg = true;
                    c4 = 1;
                }
                if (e) lists[v * 8 + e - 1].push_back(off | ((neg != eneg) ? 0x80000000u : 0u));
            }
            // c4 is 0 here: the top nibble of |d| <= 0x8000 is at most 8 with its carry
        }
        // carry is 0 here: the scalar is below 2^255, so the top digit takes it
    }
    std::vector<u32> flat, start(nlists + 1, 0);
    for (int l = 0; l < nlists; ++l) {
        start[l] = (u32)flat.size();
        flat.insert(flat.end(), lists[l].begin(), lists[l].end());
    }
    start[nlists] = (u32)flat.size();
    if (flat.empty()) flat.push_back(0);
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.collapse.reserve((size_t)34 * nJ * 128)) != H2_OK) return rc;
    if ((rc = cx.collapse_list.reserve((nlists + 1) * 4 + flat.size() * 4)) != H2_OK) return rc;
    u32 *d_start = cx.collapse_list.as<u32>(), *d_list = d_start + nlists + 1;
    // pageable sources: consumed when hipMemcpyAsync returns; stream-ordered after the previous call's kernels
    H2_HIP(hipMemcpyAsync(d_start, start.data(), (nlists + 1) * 4, hipMemcpyHostToDevice, st));
    H2_HIP(hipMemcpyAsync(d_list, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, st));
    {   // the table's columns must be complete (a registration runs on the null stream and synchronises; nothing to wait for)
        dim3 blk(256), g1((nJ + 255) / 256, 32), g2((4 * nJ + 255) / 256), g3((nJ + 255) / 256);
        u32 *sums = cx.collapse.as<u32>();
        if (!nibbles) {
            dim3 r1((nJ + 255) / 256, 16), r2((2 * nJ * kGroup + 255) / 256);
            g3 = dim3((nJ * kGroup + 255) / 256);            // combine and finish run one chain per QUAD of lanes
            if (b->curve == H2_PALLAS) {
                hipLaunchKernelGGL((ipa_readout_groups<FP>), r1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
                hipLaunchKernelGGL((ipa_readout_combine<FP>), r2, blk, 0, st, sums, nJ);
                hipLaunchKernelGGL((ipa_readout_finish<FP>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
            } else {
                hipLaunchKernelGGL((ipa_readout_groups<FQ>), r1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
                hipLaunchKernelGGL((ipa_readout_combine<FQ>), r2, blk, 0, st, sums, nJ);
                hipLaunchKernelGGL((ipa_readout_finish<FQ>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
            }
        } else if (b->curve == H2_PALLAS) {
            hipLaunchKernelGGL((ipa_collapse_buckets<FP>), g1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
            hipLaunchKernelGGL((ipa_collapse_windows<FP>), g2, blk, 0, st, sums, nJ);
            hipLaunchKernelGGL((ipa_collapse_finish<FP>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
        } else {
            hipLaunchKernelGGL((ipa_collapse_buckets<FQ>), g1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
            hipLaunchKernelGGL((ipa_collapse_windows<FQ>), g2, blk, 0, st, sums, nJ);
            hipLaunchKernelGGL((ipa_collapse_finish<FQ>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
        }
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

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
// ---- the paired commit over a SMALL 16-bit table: 8-bit sub-digits, 512 buckets, no bucket fold to speak of ----------------------
// The opening argument's rounds over the collapsed generators are paired commits of 2^14 .. 2^15 points, fifteen of them in a row at
// k = 20, each a chain of ~13 short launches through the machinery above: a two-pass sort into 2 x 2^15 buckets that hold eight
// entries each, an accumulate of four entries per lane, and a fold over 2^16 buckets (finish, two heavy-bucket launches that find
// nothing, 383 line sums per slice, 15 bit planes with a 15-doubling chain) -- 0.24 ms of which 0.05 is bucket arithmetic.  For a small
// table the same commit is cheaper with FEWER buckets: every signed 16-bit table digit d is cut once more, |d| = e_0 + 256 e_1 with e_0
// in [-127, 128] and e_1 in [0, 128] (the read-out of the collapsed generators does the same, ipa_readout_*), so that
//     sum_m c_m G_m = P_0 + 2^8 P_1,     P_pos = sum_{b < 128} (b + 1) * (sum of +-T[w][m] over the (m, w) whose sub-digit at `pos` is +-(b + 1))
// per output: 2 sides x 2 positions x 128 = 512 buckets in all, two entries per digit (2^20 entries for 2^15 + 4 scalars: the
// accumulate doubles, to the 50 us the chip needs for 2^20 mixed additions), a sort by a 9-bit key (three short launches, LDS
// histograms), a finisher in which EVERY bucket is a tree over ~256 range heads, 8 bit planes straight over each slice's 128 bucket sums
// (a slice is ONE line of the bucket matrix: no line sums) with a 7-doubling chain, and 8 doublings to join the positions.  The
// accumulate and the planes are the kernels above.  Bucket SLOTS lie 129 apart per slice (slot = key + key / 128: a slice's sums, then one
// slot that stays empty) -- the layout fold9_planes reads S column sums and NR - 1 row sums in, with S = 128 and NR = 1.
static constexpr u32 kSubKeys = 512;     // side (2) x position (2) x (|e| - 1 < 128)
static constexpr u32 kSubSlots = 516;    // 4 x 129
template <typename Fn> __device__ __forceinline__ void for_each_subdigit(const fe &s, u32 side, Fn f) {       // s: canonical, below 2^255
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const u32 raw = ((s.v[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;      // signed 16-bit digits, as msm_recode cuts them
        const bool neg = raw > 0x8000u;
        carry = neg ? 1u : 0u;
        const u32 mag = neg ? 0x10000u - raw : raw;                             // |d| <= 2^15 (raw = 2^16: digit 0, carry out)
        u32 e0 = mag & 255u, c8 = 0;
        bool e0neg = false;
        if (e0 > 128u) {
            e0 = 256u - e0;
            e0neg = true;
            c8 = 1;
        }
        const u32 e1 = (mag >> 8) + c8;                                         // <= 128: |d| <= 2^15, and |d| = 2^15 has e_0 = 0
        if (e0) f(side * 256u + e0 - 1u, (u32)w, (neg != e0neg) ? 0x80000000u : 0u);
        if (e1) f(side * 256u + 128u + e1 - 1u, (u32)w, neg ? 0x80000000u : 0u);
    }
    // (carry is 0 here: the scalar is below 2^255, the top window takes it)
}
// the same over an ENDOMORPHISM table (Bases::glv = 9): the scalar is split k = k1 + k2 lambda (glv.cuh, |k1|, |k2| < 2^129), each half cut into nine
// signed 16-bit digits -- the ninth holds bit 128 and the last carry -- and every digit into its two sub-digits; the digits of k1 read rows 0 .. 8,
// those of k2 rows 9 .. 17 (the images under phi); a negative half flips the sign of all its entries.  f(key, ROW, sign).
template <int FS, typename Fn> __device__ __forceinline__ void for_each_subdigit_glv(const fe &s, u32 side, Fn f) {
    u32 mag[2][5], hneg[2];
    glv_split<FS>(s, mag[0], hneg[0], mag[1], hneg[1]);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const bool hn = hneg[part] != 0;
        u32 carry = 0;
#pragma unroll
        for (int w = 0; w < 7; ++w) {                                                // windows 0 .. 6: signed, as in for_each_subdigit
            const u32 raw = ((mag[part][w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
            const bool neg = raw > 0x8000u;
            carry = neg ? 1u : 0u;
            const u32 m = neg ? 0x10000u - raw : raw;
            u32 e0 = m & 255u, c8 = 0;
            bool e0neg = false;
            if (e0 > 128u) {
                e0 = 256u - e0;
                e0neg = true;
                c8 = 1;
            }
            const u32 e1 = (m >> 8) + c8;
            const bool dn = neg != hn;                                                // the digit's sign times the half's
            if (e0) f(side * 256u + e0 - 1u, (u32)(part * 9 + w), (dn != e0neg) ? 0x80000000u : 0u);
            if (e1) f(side * 256u + 128u + e1 - 1u, (u32)(part * 9 + w), dn ? 0x80000000u : 0u);
        }
        // Window 7 is cut UNSIGNED (0 .. 2^16 with the carry): recoded like the others it would send a carry into window 8 for a quarter of
        // the halves, every one of those entries into the SAME bucket (position 0, magnitude 1) -- four times the average bucket, and the
        // finisher's launch is as long as its longest tree.  Its high sub-digit may then exceed 128 (up to 257): it leaves as two or three
        // entries of at most 128 each, cut evenly so that they spread over the buckets of position 1.
        {
            const u32 raw = (mag[part][3] >> 16) + carry;
            u32 e0 = raw & 255u, c8 = 0;
            bool e0neg = false;
            if (e0 > 128u) {
                e0 = 256u - e0;
                e0neg = true;
                c8 = 1;
            }
            u32 e1 = (raw >> 8) + c8;
            if (e0) f(side * 256u + e0 - 1u, (u32)(part * 9 + 7), (hn != e0neg) ? 0x80000000u : 0u);
            // (split EVENLY: "128 and the rest" would pile a quarter of the halves into the one bucket of magnitude 128)
            const u32 pieces = (e1 + 127u) / 128u;                  // 0 .. 3
            for (u32 j = 0; j < pieces; ++j) {
                const u32 d = (e1 + j) / pieces;                    // floor((e1 + j) / pieces), j < pieces: sums to e1, each <= 128
                f(side * 256u + 128u + d - 1u, (u32)(part * 9 + 7), hn ? 0x80000000u : 0u);
            }
        }
        // window 8: bit 128 and beyond (glv_split promises < 2^129; nothing has been seen above 2^128) -- unsigned too, almost always nothing
        {
            u32 rest = min(mag[part][4], 256u);        // (<= 1 by glv_split's bound, which tests/test_glv_constants.py holds the constants to; the clamp keeps a
                                                       // broken promise from running past the entry list: two entries per half are reserved for this window)
            while (rest) {
                const u32 d = rest > 128u ? 128u : rest;
                rest -= d;
                f(side * 256u + d - 1u, (u32)(part * 9 + 8), hn ? 0x80000000u : 0u);
            }
        }
    }
}
__device__ __forceinline__ u32 pair_side(u32 i, u32 pair_n, int pair_shift) { return i < pair_n ? (i >> pair_shift) & 1u : (i - pair_n) & 1u; }
// pass A: a workgroup's 512 scalars -> its 512 counters (H2_SUB_BLOCK = 256 / 512 / 1024, late round of the k = 20 argument: 0.220 / 0.219 / 0.227 ms --
// fewer workgroups shorten pass B's walk, more of them pass C's scattered stores)
static constexpr u32 kSubBlock = 512;
template <int FS>
__global__ void __launch_bounds__(1024) sub_count(const u32 *__restrict__ scalars, u32 n, u32 pair_n, int pair_shift, int mont, int glv,
                                                       u32 *__restrict__ wg_hist) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kSubKeys];
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        fe s = fe_load(scalars + 8 * (size_t)i);
        if (mont) s = fe_redc<FS>(s);
        if (glv) for_each_subdigit_glv<FS>(s, pair_side(i, pair_n, pair_shift), [&](u32 key, u32, u32) { atomicAdd(&sh[key], 1u); });
        else for_each_subdigit(s, pair_side(i, pair_n, pair_shift), [&](u32 key, u32, u32) { atomicAdd(&sh[key], 1u); });
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) wg_hist[(size_t)blockIdx.x * kSubKeys + k] = sh[k];
}
// pass B (one workgroup, a lane per key): the workgroups' offsets inside each key's run, every key's start (`kstart`, for pass C), the boundary
// array over the 516 SLOTS (+ total + the sentinel msm_accumulate reads; a gap slot is an empty bucket), and the raw bucket slots cleared for the
// accumulate (36 words each: a lane clears its own, lanes 0 .. 3 the gaps too)
__global__ void __launch_bounds__(kSubKeys) sub_scan(const u32 *__restrict__ wg_hist, u32 nblk, u32 *__restrict__ wg_off, u32 *__restrict__ kstart,
                                                     u32 *__restrict__ starts, u32 *__restrict__ buckets9) {
    H2_LATENCY_STAGE();
    __shared__ u32 tot[kSubKeys];
    const u32 k = threadIdx.x;
    u32 run = 0;
    for (u32 b0 = 0; b0 < nblk; b0 += 8) {               // eight loads in flight: the walk is a chain of memory round trips otherwise
        u32 c[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) c[j] = b0 + j < nblk ? wg_hist[(size_t)(b0 + j) * kSubKeys + k] : 0u;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
            if (b0 + j < nblk) wg_off[(size_t)(b0 + j) * kSubKeys + k] = run;
            run += c[j];
        }
    }
    const u32 slot = k + (k >> 7);
#pragma unroll
    for (int i = 0; i < 36; ++i) buckets9[36 * (size_t)slot + i] = 0u;
    if (k < 4)
        for (int i = 0; i < 36; ++i) buckets9[36 * (size_t)(129 * k + 128) + i] = 0u;
    tot[k] = run;
    __syncthreads();
    for (u32 off = 1; off < kSubKeys; off <<= 1) {
        const u32 t = k >= off ? tot[k - off] : 0u;
        __syncthreads();
        tot[k] += t;
        __syncthreads();
    }
    kstart[k] = tot[k] - run;
    starts[slot] = tot[k] - run;
    if ((k & 127u) == 127u) starts[slot + 1] = tot[k];          // the gap behind a slice: starts where the next slice starts
    if (k == kSubKeys - 1) {
        starts[kSubSlots] = tot[k];
        starts[kSubSlots + 1] = 0xFFFFFFFFu;
    }
}
// pass C: the same digits again, each to its place (entry = table index | sign << 31; the order inside a bucket is immaterial)
template <int FS>
__global__ void __launch_bounds__(1024) sub_scatter(const u32 *__restrict__ scalars, u32 n, u32 pair_n, int pair_shift, int mont, int glv, u32 stride,
                                                   const u32 *__restrict__ wg_off, const u32 *__restrict__ kstart, u32 *__restrict__ entries) {
    H2_LATENCY_STAGE();
    __shared__ u32 cur[kSubKeys];
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) cur[k] = kstart[k] + wg_off[(size_t)blockIdx.x * kSubKeys + k];
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe s = fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_redc<FS>(s);
    auto place = [&](u32 key, u32 row, u32 sign) {
        const u32 pos = atomicAdd(&cur[key], 1u);
        entries[pos] = (row * stride + i) | sign;
    };
    if (glv) for_each_subdigit_glv<FS>(s, pair_side(i, pair_n, pair_shift), place);
    else for_each_subdigit(s, pair_side(i, pair_n, pair_shift), place);
}
// the finisher when EVERY bucket owns hundreds of range heads: a workgroup per bucket, a tree over its heads, then the bucket's own segment
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_finish_dense(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                           u32 total_buckets, u32 T, u32 div) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    const u32 b = blockIdx.x, base = starts[0];
    const u32 M = starts[total_buckets] - base;
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    const u32 h0 = (starts[b] - base + chunk - 1) / chunk, h1 = (starts[b + 1] - base + chunk - 1) / chunk;
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(heads9, h1 > h0 ? h1 - h0 : 0u, [h0](u32 k) { return h0 + k; });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (!fold9_root()) return;
    xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(buckets9 + 36 * (size_t)b));
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz9_store_raw<FB>(buckets9 + 36 * (size_t)b, acc);
}
// The rest of the fold of the four 128-bucket slices in ONE launch (fold9_planes' scheme, without line sums and with the join of the two
// positions inside): workgroup (t, y) sums plane t of slice y = side * 2 + pos -- the 64 finished buckets with bit t of b + 1 set (plane 7:
// bucket 127 alone) -- and doubles it t times; the workgroup that arrives LAST at its slice adds the eight planes (three tree levels), doubles
// the sum eight more times when the slice is a position 1 (its buckets count in units of 2^8), and of the two slices of a side the one that
// arrives last adds the other's sum and writes output `side`.  Arrival counters behind fences, left at zero: counter[0..3] the slices',
// counter[4..5] the sides'.
template <int FB>
__global__ void __launch_bounds__(256, 3) sub_planes(const u32 *__restrict__ buckets9, u32 *__restrict__ planes9, u32 *__restrict__ sums9,
                                                   u32 *__restrict__ counter, u32 *__restrict__ out, int out_kind, int out_mont) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    __shared__ u32 s_last;
    const u32 t = blockIdx.x, y = blockIdx.y;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 *src = buckets9 + 36 * (size_t)129 * y;
    planes9 += 36 * (size_t)8 * y;
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(src, t < 7 ? 64u : 1u, [t](u32 k) {
        return t < 7 ? ((((k >> t) << (t + 1)) | (1u << t) | (k & ((1u << t) - 1u))) - 1u) : 127u;      // the k-th b with bit t of b + 1 set
    });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (fold9_root()) {
        for (u32 k = 0; k < t; ++k) acc = xyzz9_dbl_wide<FB>(acc);
        if (lead) {
            xyzz9_store_raw<FB>(planes9 + 36 * (size_t)t, acc);
            __threadfence();
            s_last = atomicAdd(counter + y, 1u) == 7u ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    acc = fold9_quad_gather<FB, H2_FOLD_D>(planes9, 8u, [](u32 k) { return k; });
    acc = fold9_quads_sum<FB>(acc, sh, 8u);
    if (!fold9_root()) return;
    if (y & 1u)
        for (int k = 0; k < 8; ++k) acc = xyzz9_dbl_wide<FB>(acc);
    u32 ticket = 0;
    if (lead) {
        counter[y] = 0;
        xyzz9_store_raw<FB>(sums9 + 36 * (size_t)y, acc);
        __threadfence();
        ticket = atomicAdd(counter + 4 + (y >> 1), 1u);
    }
    ticket = (u32)__builtin_amdgcn_mov_dpp((int)ticket, 0, 0xf, 0xf, false);      // quad lane 0's ticket
    if (ticket == 0) return;                                                       // the side's other slice finishes the output
    __threadfence();
    xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(sums9 + 36 * (size_t)(y ^ 1u)));
    const xyzz<FB> r = xyzz9_to_r256_wide<FB>(acc);
    if (!lead) return;
    counter[4 + (y >> 1)] = 0;
    out += (out_kind == H2_OUT_AFFINE ? 16 : 24) * (size_t)(y >> 1);
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

template <int FB, int FS>
static int pair_subdigit_launch(MsmContext &cx, const Bases &b, const void *d_scalars, size_t n, unsigned pair_shift, int form, int out_kind,
                                void *d_out, hipStream_t st) {
    int rc;
    static const u32 sub_block = [] { const char *e = getenv("H2_SUB_BLOCK"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : kSubBlock); }();   // A/B
    const u32 nblk = (u32)((n + sub_block - 1) / sub_block), tb = kSubSlots, nsl = 4;
    const int glv = b.glv ? 1 : 0;
    const size_t max_entries = n * (glv ? 40 : 32);             // two sub-digits per table digit: 16 digits; over an endomorphism table 2 x (7 x 2 + 4 + 2) at most
    u32 &lanes = cx.lanes[FB][2];
    if (!lanes) {            // how many lanes of the M9 accumulate the chip holds at once (as msm_launch sizes it)
        int dev = 0, cus = 0, per_cu = 0;
        H2_HIP(hipGetDevice(&dev));
        H2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, false, true>, 256, 0));
        per_cu = std::min(per_cu, (int)H2_ACC9_WAVES);
        lanes = (u32)cus * (u32)std::max(per_cu, 1) * 256u;
    }
    // entries per lane: 8 over a plain table (2^14 + 4 scalars: 2^19 entries, 128 range heads per bucket -- exactly what the finisher's 64 quads take in
    // ONE round of two gathers each); an endomorphism table leaves ~33 entries per scalar, which at 8 per lane is 131 heads per bucket and a second
    // round for three of them (finisher 35 -> 54 us): 9 per lane there
    const u32 lane_div = glv ? 9 : 8;
    const u32 T = (u32)std::min<size_t>(lanes, std::max<size_t>(256, (max_entries / lane_div + 255) / 256 * 256));
    if ((rc = cx.hist.reserve(((size_t)2 * nblk + 1) * kSubKeys * 4)) != H2_OK || (rc = cx.starts.reserve((tb + 2) * 4)) != H2_OK ||
        (rc = cx.entries.reserve(max_entries * 4)) != H2_OK || (rc = cx.seg9.reserve(((size_t)T + tb) * 144)) != H2_OK ||
        (rc = cx.partial.reserve((size_t)nsl * (8 + 1) * 144)) != H2_OK)
        return rc;
    if (cx.fold_ctr.cap < 64) {          // fold9_planes' arrival counters: zero once, every launch leaves them at zero
        if ((rc = cx.fold_ctr.reserve((size_t)kMaxCols * 64)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.fold_ctr.ptr, 0, (size_t)kMaxCols * 64, st));
    }
    u32 *wg_hist = cx.hist.as<u32>(), *wg_off = wg_hist + (size_t)nblk * kSubKeys, *kstart = wg_off + (size_t)nblk * kSubKeys;
    u32 *starts = cx.starts.as<u32>(), *entries = cx.entries.as<u32>();
    u32 *heads9 = cx.seg9.as<u32>(), *buckets9 = heads9 + 36 * (size_t)T;
    u32 *planes9 = cx.partial.as<u32>(), *sums9 = planes9 + 36 * (size_t)nsl * 8;
    const int mont = form == H2_FORM_MONTGOMERY ? 1 : 0;
    const u32 pair_n = (u32)(n - 4);
    ColStride cs;
    memset(&cs, 0, sizeof cs);
    hipLaunchKernelGGL((sub_count<FS>), dim3(nblk), dim3(sub_block), 0, st, (const u32 *)d_scalars, (u32)n, pair_n, (int)pair_shift, mont, glv, wg_hist);
    hipLaunchKernelGGL(sub_scan, dim3(1), dim3(kSubKeys), 0, st, (const u32 *)wg_hist, nblk, wg_off, kstart, starts, buckets9);
    hipLaunchKernelGGL((sub_scatter<FS>), dim3(nblk), dim3(sub_block), 0, st, (const u32 *)d_scalars, (u32)n, pair_n, (int)pair_shift, mont, glv, b.stride,
                       (const u32 *)wg_off, (const u32 *)kstart, entries);
    hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(T / 256), dim3(256), 0, st, (const u32 *)b.d_table, (const u32 *)nullptr, 0xFFFFFFFFu,
                       (const u32 *)entries, (const u32 *)starts, heads9, buckets9, tb, T, lane_div, cs);
    hipLaunchKernelGGL((fold9_finish_dense<FB>), dim3(tb), dim3(256), 0, st, (const u32 *)heads9, (const u32 *)starts, buckets9, tb, T, lane_div);
    hipLaunchKernelGGL((sub_planes<FB>), dim3(8, nsl), dim3(256), 0, st, (const u32 *)buckets9, planes9, sums9, cx.fold_ctr.as<u32>(), (u32 *)d_out, out_kind,
                       mont);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_commit_pair_device(h2_bases_t g, const void *d_scalars, size_t n, unsigned pair_shift, int form, int out_kind,
                                     void *d_out, void *stream) {
    auto b = find_bases(g, true);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !d_out || !d_scalars || n != b->n || n < 8 || pair_shift > 31) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    // small 16-bit tables (the opening argument's rounds over the collapsed generators): the 8-bit sub-digit form above.
    // H2_PAIR_SUBDIGITS=0: the general form for every size (A/B); = n: the largest table (points) that takes the sub-digit form.
    if (b->glv || (pair_subdigits_apply(n) && b->c == 16 && b->W == 16 && !prof_enabled() && !timeline_on())) {      // (an endomorphism table has no other reader)
        if (b->curve == H2_PALLAS) return pair_subdigit_launch<FP, FQ>(cx, *b, d_scalars, n, pair_shift, form, out_kind, d_out, st);
        return pair_subdigit_launch<FQ, FP>(cx, *b, d_scalars, n, pair_shift, form, out_kind, d_out, st);
    }
    MsmArgs a{d_scalars, nullptr, b->d_table, nullptr, n, true, b->c, b->stride, (u32)b->n, form, out_kind, d_out};
    a.pair_shift = (int)pair_shift;
    a.pair_n = (u32)(n - 4);
    return msm_dispatch(cx, b->curve, a, st);
}

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
    static const int batch_env = [] { const char *e = getenv("H2_BATCH_COLS"); int v = e ? atoi(e) : 0; return v >= 1 && v <= kMaxCols ? v : 0; }();
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

// ---- Params::commit from a HOST column (the literal seam: the reference's `poly` is a Vec in host memory,
// poly/commitment.rs:119-130).  The column is cut into ranges of the registered table's columns; range r is copied on a copy
// stream and, as soon as it has landed, committed on one of two compute streams as a multiexp of its own over table columns
// [lo, hi) (MsmArgs::col0) -- so PCIe runs beside the bucket arithmetic of the ranges before it, and only the first range's
// copy and ONE fold stay exposed: a range stops after its buckets are finished and adds them into a running bucket slice
// (MsmArgs::add_into, one per compute stream); the summed slice is folded once (MsmArgs::fold_from).
namespace {
constexpr int kPipeMaxChunks = 16;
constexpr size_t kPipeChunk = (size_t)1 << 18;      // 8 MiB of scalars: 0.15 ms of PCIe, 0.26 ms of bucket additions.  Measured at 2^20 (bench/tools/
                                                    // host_commit_sweep.py): ranges of 2^20 / 2^19 / 2^18 / 2^17 / 2^16 -> 2.25 / 1.88 / 1.80 / 2.20 / 3.10 ms
                                                    // (every range pays its own sort chain and bucket finish); resident commit 1.52, raw copy 0.59
struct HostPipe {
    std::mutex mu;
    bool ready = false;
    hipStream_t copy = nullptr, comp[2] = {nullptr, nullptr};
    hipEvent_t landed[kPipeMaxChunks] = {nullptr}, done[2] = {nullptr, nullptr};
    DevBuf stage, parts;
    int prepare() {
        if (ready) return H2_OK;
        H2_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        for (auto &c : comp) H2_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        for (auto &e : landed) H2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : done) H2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ready = true;
        return H2_OK;
    }
};
HostPipe &host_pipe() {
    static HostPipe p[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return p[dev & 15];
}
}  // namespace


namespace h2 {
void msm_release_host_pipe() {            // h2_trim: the staging column and the running bucket slices of h2_commit (the device is idle)
    HostPipe &hp = host_pipe();
    std::lock_guard<std::mutex> lk(hp.mu);
    hp.stage.release();
    hp.parts.release();
}
}  // namespace h2

static int commit_host_pipelined_locked(HostPipe &hp, Bases &b, const uint64_t *scalars, size_t n, const uint64_t *blind, int form, int out_kind,
                                       uint64_t *out);
// The pipeline lives on the device that holds the table: the current device is switched to it for the call (the host pipe, its
// streams and the msm contexts are all keyed by the current device), and ANY early return -- a HIP error half-way through the
// ranges -- first waits for the copy stream and both compute streams, so that nothing still runs against hp.stage / hp.parts when
// the next call reserves (hipFree) or zeroes them.
static int commit_host_pipelined(Bases &b, const uint64_t *scalars, size_t n, const uint64_t *blind, int form, int out_kind, uint64_t *out) {
    int cur = 0;
    H2_HIP(hipGetDevice(&cur));
    if (cur != b.device) H2_HIP(hipSetDevice(b.device));
    int rc;
    {
        HostPipe &hp = host_pipe();
        std::lock_guard<std::mutex> lk(hp.mu);
        rc = hp.prepare();
        if (rc == H2_OK) rc = commit_host_pipelined_locked(hp, b, scalars, n, blind, form, out_kind, out);
        if (rc != H2_OK && hp.ready) {
            (void)hipStreamSynchronize(hp.copy);
            for (auto &c : hp.comp) (void)hipStreamSynchronize(c);
        }
    }
    if (cur != b.device) (void)hipSetDevice(cur);
    return rc;
}
static int commit_host_pipelined_locked(HostPipe &hp, Bases &b, const uint64_t *scalars, size_t n, const uint64_t *blind, int form, int out_kind,
                                       uint64_t *out) {
    int rc = H2_OK;
    size_t chunk = g_pipe_chunk.load() ? g_pipe_chunk.load() : kPipeChunk;
    if (n < 2 * chunk) chunk = std::max<size_t>(n, 1);                       // small columns: one range, nothing to overlap
    chunk = std::max(chunk, (n + kPipeMaxChunks - 1) / kPipeMaxChunks);
    const int chunks = (int)std::max<size_t>(1, (n + chunk - 1) / chunk);
    const size_t nb = (size_t)1 << (b.c - 1);
    if ((rc = hp.stage.reserve(n * 32 + 64)) != H2_OK) return rc;
    if ((rc = hp.parts.reserve(2 * nb * 128 + 128)) != H2_OK) return rc;
    char *d_s = hp.stage.as<char>(), *d_blind = d_s + n * 32;
    u32 *total[2] = {hp.parts.as<u32>(), hp.parts.as<u32>() + 32 * nb};
    char *d_res = (char *)(total[1] + 32 * nb);
    const int used = chunks > 1 ? 2 : 1;                                     // compute streams in play
    for (int j = 0; j < used; ++j) H2_HIP(hipMemsetAsync(total[j], 0, nb * 128, hp.comp[j]));
    if (blind) H2_HIP(hipMemcpyAsync(d_blind, blind, 32, hipMemcpyHostToDevice, hp.copy));
    for (int i = 0; i < chunks; ++i) {
        const size_t lo = (size_t)i * chunk, len = std::min(chunk, n - lo);
        if (len) H2_HIP(hipMemcpyAsync(d_s + lo * 32, (const char *)scalars + lo * 32, len * 32, hipMemcpyHostToDevice, hp.copy));
        H2_HIP(hipEventRecord(hp.landed[i], hp.copy));
        hipStream_t st = hp.comp[i & 1];
        H2_HIP(hipStreamWaitEvent(st, hp.landed[i], 0));
        MsmContext &cx = msm_ctx(st);
        std::lock_guard<std::mutex> cl(cx.mu);
        // the blind rides with the last range (its base is column n of the table whatever the range)
        MsmArgs a{d_s + lo * 32, (blind && i == chunks - 1) ? d_blind : nullptr, b.d_table, nullptr, len, true, b.c, b.stride, (u32)b.n, form,
                  H2_OUT_JACOBIAN, nullptr};
        a.col0 = (u32)lo;
        a.add_into = total[i & 1];
        if ((rc = msm_dispatch(cx, b.curve, a, st)) != H2_OK) break;
    }
    for (int j = 0; j < used; ++j) {
        H2_HIP(hipEventRecord(hp.done[j], hp.comp[j]));
        H2_HIP(hipStreamWaitEvent(hp.copy, hp.done[j], 0));
    }
    if (rc == H2_OK) {
        if (used == 2) {
            const dim3 grid((unsigned)((nb * kGroup + 255) / 256)), blk(256);
            if (b.curve == H2_PALLAS) hipLaunchKernelGGL((msm_bucket_add<FP>), grid, blk, 0, hp.copy, total[0], (const u32 *)total[1], (u32)nb);
            else hipLaunchKernelGGL((msm_bucket_add<FQ>), grid, blk, 0, hp.copy, total[0], (const u32 *)total[1], (u32)nb);
        }
        MsmContext &cx = msm_ctx(hp.copy);
        std::lock_guard<std::mutex> cl(cx.mu);
        MsmArgs a{nullptr, nullptr, b.d_table, nullptr, 0, true, b.c, b.stride, (u32)b.n, form, out_kind, d_res};
        a.fold_from = total[0];
        rc = msm_dispatch(cx, b.curve, a, hp.copy);
    }
    if (rc != H2_OK) return rc;                      // (the caller drains the three streams)
    H2_HIP(hipMemcpyAsync(out, d_res, out_kind == H2_OUT_AFFINE ? 64 : 96, hipMemcpyDeviceToHost, hp.copy));
    H2_HIP(hipStreamSynchronize(hp.copy));
    return H2_OK;
}

extern "C" int h2_commit(h2_bases_t g, const uint64_t *scalars, size_t n, const uint64_t *w_xy, const uint64_t *blind,
                         int form, int out_kind, uint64_t *out) {
    auto b = find_bases(g);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !out || (n && !scalars) || n > b->n || (w_xy && !blind)) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (blind) {   // w_xy: compared by content with the handle's blind base, installed if it differs; NULL: the handle's own
        if (w_xy) {
            if ((rc = set_blind_base_host(*b, w_xy, form)) != H2_OK) return rc;
        } else {
            std::lock_guard<std::mutex> bl(b->mu);
            if (!b->blind_set) {
                set_last_error_msg("h2_commit with a blind but the handle has no blind base: call h2_bases_set_blind_base, or pass w_xy");
                return H2_ERR_ARGS;
            }
        }
    }
    return commit_host_pipelined(*b, scalars, n, blind, form, out_kind, out);
}

// device-resident partials (the landing buffer of an all-gather) -> their sum, on `stream`
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
