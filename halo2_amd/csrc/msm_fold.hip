// After the accumulate: finish (range heads into their buckets), the fold of a bucket slice down to its weighted sum, and the Horner step over
// window slices (reference: the running-sum fold of arithmetic.rs:86-92 and the window loop of :169-178, restructured as trees).
#include "msm_internal.cuh"

namespace h2 {

// ---- finisher: bucket b = its own non-head segment + the heads of the ranges that begin inside it -----
// Buckets owning more than kHeavy heads (scalars repeated thousands of times) are parked on a list and summed
// by a whole workgroup each in msm_finish_heavy.
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
    if (!cs.lowprio) H2_LATENCY_STAGE();        // (a fold with slack -- a group of a generic multiexp folded beside the next group's accumulate -- keeps priority 0)
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
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_finish_heavy(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs) {
    if (!cs.lowprio) H2_LATENCY_STAGE();        // (a fold with slack -- a group of a generic multiexp folded beside the next group's accumulate -- keeps priority 0)
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
    if (!cs.lowprio) H2_LATENCY_STAGE();        // (a fold with slack -- a group of a generic multiexp folded beside the next group's accumulate -- keeps priority 0)
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
    if (!cs.lowprio) H2_LATENCY_STAGE();        // (a fold with slack -- a group of a generic multiexp folded beside the next group's accumulate -- keeps priority 0)
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
    if (!cs.lowprio) H2_LATENCY_STAGE();        // (a fold with slack -- a group of a generic multiexp folded beside the next group's accumulate -- keeps priority 0)
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
// call then adds that point (`addend`) to its own Horner sum and emits the result.  addend_first (the grouped form, msm_generic.hip): the
// addend -- the chain of the groups above, already doubled down to the weight of this group's LOWEST slice -- is added BEFORE the extra
// doublings that carry the sum on to the next group: A_g = R_g + D_(g-1), D_g = 2^(extra) A_g.
template <int FB>
__global__ void __launch_bounds__(64) msm_combine(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl, const u32 *__restrict__ addend, int addend_first) {
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
    if (addend && addend_first) xyzz9_add_wide<FB>(r9, xyzz9_from_r256_wide<FB>(xyzz_load<FB>(addend)));
    for (int k = 0; k < extra_dbl; ++k) r9 = xyzz9_dbl_wide<FB>(r9);
    if (addend && !addend_first) xyzz9_add_wide<FB>(r9, xyzz9_from_r256_wide<FB>(xyzz_load<FB>(addend)));
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


// ---- explicit instantiations (both curves) ----
template __global__ void msm_finish_buckets<FP>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div);
template __global__ void msm_finish_buckets<FQ>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                          u32 *__restrict__ buckets, u32 *__restrict__ heavy,
                                                          u32 total_buckets, u32 T, u32 div);
template __global__ void msm_finish_heavy<FP>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div);
template __global__ void msm_finish_heavy<FQ>(const u32 *__restrict__ heads, const u32 *__restrict__ starts,
                                                        u32 *__restrict__ scratch, const u32 *__restrict__ heavy,
                                                        u32 total_buckets, u32 T, u32 div);
template __global__ void msm_finish_heavy2<FP>(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy);
template __global__ void msm_finish_heavy2<FQ>(const u32 *__restrict__ scratch, u32 *__restrict__ buckets,
                                                        const u32 *__restrict__ heavy);
template __global__ void msm_bucket_add<FP>(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb);
template __global__ void msm_bucket_add<FQ>(u32 *__restrict__ total, const u32 *__restrict__ part, u32 nb);
template __global__ void fold9_finish<FP>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void fold9_finish<FQ>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                    u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void fold9_finish_heavy<FP>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void fold9_finish_heavy<FQ>(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ scratch9,
                                                          const u32 *__restrict__ heavy, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void fold9_finish_heavy2<FP>(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs);
template __global__ void fold9_finish_heavy2<FQ>(const u32 *__restrict__ scratch9, u32 *__restrict__ buckets9, const u32 *__restrict__ heavy,
                                                             ColStride cs);
template __global__ void fold9_rowcol<FP>(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs);
template __global__ void fold9_rowcol<FQ>(const u32 *__restrict__ buckets9, u32 *__restrict__ lines9, u32 S, u32 NR, ColStride cs);
template __global__ void fold9_planes<FP>(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs);
template __global__ void fold9_planes<FQ>(const u32 *__restrict__ lines9, u32 *__restrict__ planes9, u32 *__restrict__ counter, u32 S, u32 NR,
                                                    int cb, u32 *__restrict__ out, int out_kind, int out_mont, ColOut co, ColStride cs);
template __global__ void msm_reduce_segments<FP>(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg);
template __global__ void msm_reduce_segments<FQ>(const u32 *__restrict__ buckets, u32 *__restrict__ partial,
                                                           u32 NB, u32 total_segments, int seg);
template __global__ void msm_sum_slice<FP>(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share);
template __global__ void msm_sum_slice<FQ>(const u32 *__restrict__ partial, u32 *__restrict__ out, u32 per_slice,
                                                     u32 share);
template __global__ void msm_rowcol_sums<FP>(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR);
template __global__ void msm_rowcol_sums<FQ>(const u32 *__restrict__ buckets, u32 *__restrict__ wide, u32 S, u32 NR);
template __global__ void msm_combine<FP>(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl, const u32 *__restrict__ addend, int addend_first);
template __global__ void msm_combine<FQ>(const u32 *__restrict__ slice_sums, int slices, int c, u32 *__restrict__ out,
                                                  int out_kind, int out_mont, int extra_dbl, const u32 *__restrict__ addend, int addend_first);
template __global__ void msm_combine_ranges<FP>(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont);
template __global__ void msm_combine_ranges<FQ>(RangeSums rs, int ranges, int slices, int c, u32 *__restrict__ out, int out_kind, int out_mont);

}  // namespace h2
