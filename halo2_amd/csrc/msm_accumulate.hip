// The hot kernel of the multiexp: bucket accumulation over the sorted entry list (msm_launch.hip describes the stages around it).
#include "msm_internal.cuh"

namespace h2 {

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
template <int FB, bool GLV, bool M9, int BLOCK>
__global__ void __launch_bounds__(BLOCK, (M9 ? H2_ACC9_WAVES * 256 / BLOCK : 4)) msm_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
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


// ---- explicit instantiations (both curves) ----
template __global__ void msm_bases_to_m9_glv<FP>(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n);
template __global__ void msm_bases_to_m9_glv<FQ>(const u32 *__restrict__ bases, u32 *__restrict__ out, u32 n);
template __global__ void msm_accumulate<FP, false, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FP, true, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FP, false, true>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FQ, false, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FQ, true, false>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FQ, false, true>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FP, false, true, 512>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_accumulate<FQ, false, true, 512>(const u32 *__restrict__ bases, const u32 *__restrict__ extra_base,
                                                      u32 extra_index, const u32 *__restrict__ entries,
                                                      const u32 *__restrict__ starts, u32 *__restrict__ heads,
                                                      u32 *__restrict__ buckets, u32 total_buckets, u32 T, u32 div, ColStride cs);
template __global__ void msm_segments_to_r256<FP>(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets);
template __global__ void msm_segments_to_r256<FQ>(const u32 *__restrict__ raw, u32 *__restrict__ heads,
                                                            u32 *__restrict__ buckets, u32 T, u32 total_buckets);

}  // namespace h2
