// best_multiexp(coeffs, bases) without a registered table (halo2_proofs/src/arithmetic.rs:143-180, benches/msm.rs:14-22), large sizes:
// the GROUPED form.
//
// The generic multiexp splits every scalar k = k1 + k2 lambda (glv.cuh) into two half-length digit columns -- 2n columns over nine window
// slices of 2^15 buckets at 16-bit windows -- converts the caller's bases once per call to M9 form (+ phi(P)), and runs the same
// accumulate and fold kernels as a registered commit, followed by a Horner step over the window slices (128 dependent doublings).
// Round 5 ran ONE sort over all slices in front of the first addition (0.25 ms at 2^20, 0.59 / 1.37 ms at 2^21 / 2^22, where its bins
// outgrew LDS) with the endomorphism split done in both of its passes, and everything behind the last addition -- the lower slices'
// fold and the end of the chain -- serial.  Here:
//
//   * the split happens ONCE (msm_glv_digits): one row of 16-bit digit codes per window slice;
//   * the window slices are cut into GROUPS, upper slices first (a lone call: 5 + 2 + 2 of nine).  Each group has its own two-pass sort (pass 1
//     reads the group's digit rows, 2 bytes per entry; pass 2 a workgroup per bin in LDS at every size), its own sorted list and
//     boundary array, its own accumulate launch, its own fold;
//   * only the FIRST group's sort stands in front of the first addition: the others are sorted beside the first accumulate (the sort
//     kernels are light now: 8-24 registers, no field arithmetic), and every group but the last is folded beside the next group's accumulate;
//   * the Horner chain is cut at the group boundaries: group g's link A_g = R_g + D_(g-1), D_g = 2^(c ns_(g+1)) A_g runs behind that
//     group's fold, so behind the last addition stand only the last (smallest) group's fold, its c (ns - 1) doublings and one addition;
//   * TWO FORMS.  Latency form (a lone call): three groups on three streams of the library's own, created back to back -- accumulates on one,
//     everything beside them on the other two -- because HIP multiplexes streams onto four hardware queues and a side stream sharing the
//     CALLER's queue would run behind the accumulate it was meant to run beside; the caller's stream only waits for the result.  8 CUs are left
//     without an accumulate workgroup and the chain links are fenced onto them by LDS requests (a chain link is one wave holding a SIMD).
//     Throughput form (another stream's generic multiexp is in flight, below 2^22 points): ONE group, everything on the caller's stream.
//
// Measured against round 5's form on one box (profiles/r06_generic_grouped.txt): 2^20 one call 1.50-1.58 against 1.55-1.62 ms, three streams
// 1.16-1.20 against 1.33-1.43; 2^21 2.59-2.62 against 2.84-2.87; 2^22 4.98-5.13 against 5.68-5.74 ms.  DESIGN.md section 4.4 has the reasons.
//
// Results are the same group element as every other form (the order of additions inside a bucket is free: SURVEY.md appendix A.1);
// parity (bit-exact canonical affine coordinates against the C restatement of the reference): tests/test_gpu_generic_grouped.py, tests/test_gpu_parity.py, build/h2bench msm / parity.
#include "msm_internal.cuh"

namespace h2 {

namespace {

struct Group {
    u32 w0 = 0, ns = 0;     // window slices [w0, w0 + ns)
    u32 tb = 0;             // its buckets: ns * NB
    GroupSort gs;           // pass 1
    Sort2 p2;               // what the pass-2 kernels read (lowb, lb, nh; no side array)
    size_t cap = 0;         // pass-2 stage of a bin's workgroup, in entries
    size_t emax = 0;        // most entries it can hold: ns * digit columns
    size_t plan_words = 0;
    u32 B1 = 0;             // pass-1 workgroups
    u32 T = 0;              // lanes of its accumulate
    size_t ent_off = 0;     // its sorted list inside cx.entries (entries)
    size_t starts_off = 0;  // its boundary array inside cx.starts (words): tb + 2 of them
    size_t bucket_off = 0;  // its first bucket slot
};

// Pass-1 bins such that an average bin is ~16 K entries (what a pass-2 workgroup stages in LDS with room to spare), the low key bits
// riding in the tagged entry above the column index.
bool group_geometry(u32 cols, u32 nb, Group &g) {
    int lb = 0;
    while (((u64)(cols - 1) >> lb) != 0) ++lb;
    g.tb = g.ns * nb;
    g.emax = (size_t)g.ns * cols;
    for (int lowb = 12; lowb >= 1; --lowb) {
        if (lowb + lb > 31) continue;
        const size_t nbk = (size_t)1 << lowb;
        const u32 nh = (u32)(((size_t)g.tb + nbk - 1) >> lowb);
        if (nh > 4096) break;
        const size_t avg = g.emax / nh;
        const size_t cap_max = nbk * 8 + 64 < kLdsCap ? (kLdsCap - nbk * 8) / 4 : 0;
        const size_t cap = std::min(cap_max, std::max<size_t>(4096, avg * 5 / 4 + 1024));
        if (avg > 20000 || avg > cap * 9 / 10) continue;
        // digit columns per pass-1 workgroup: 4096 (two workgroups per CU) -- 8192 where that still leaves >= 512 workgroups: every workgroup
        // writes and re-reads a histogram row of nh words and copies its entries out in runs of S ns / nh, so large problems want large S
        u32 S = 4096;
        while (S > 512 && ((size_t)nh * 3 + 1 + (size_t)S * g.ns) * 4 > kLdsCap / 2) S /= 2;
        if (S == 4096 && cols / 8192 >= 512 && ((size_t)nh * 3 + 1 + (size_t)8192 * g.ns) * 4 <= kLdsCap) S = 8192;
        if (((size_t)nh * 3 + 1 + (size_t)S * g.ns) * 4 > kLdsCap) continue;
        memset(&g.gs, 0, sizeof g.gs);
        g.gs.cols = cols;
        g.gs.row = (cols + 7) & ~7u;
        g.gs.w0 = g.w0;
        g.gs.ns = g.ns;
        g.gs.nb = nb;
        g.gs.lowb = lowb;
        g.gs.lb = lb;
        g.gs.nh = nh;
        g.gs.S = S;
        g.gs.run_lanes = 16;
        memset(&g.p2, 0, sizeof g.p2);
        g.p2.lowb = lowb;
        g.p2.lb = lb;
        g.p2.nh = nh;
        g.p2.nb = nb;
        g.p2.pair_shift = -1;
        g.p2.extra_col = 0xFFFFFFFFu;
        g.cap = cap;
        g.B1 = (cols + S - 1) / S;
        g.plan_words = ((size_t)nh * 2 + 1 + 64 + (((size_t)kMaxBig * (kBigChunks + 1)) << lowb) + 3) & ~(size_t)3;
        return true;
    }
    return false;
}

}  // namespace

template <int FB, int FS>
int msm_generic_grouped(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st) {
    static const bool on = [] { const char *e = ab_env("H2_GENERIC_GROUPED"); return !(e && atoi(e) == 0); }();       // 0: round 5's slice split (A/B)
    static const size_t min_n = [] { const char *e = ab_env("H2_GENERIC_GROUPED_MIN"); return e ? (size_t)atol(e) : ((size_t)1 << 18) + 1; }();
    if (!on || scalars_n < min_n || scalars_n > ((size_t)1 << 24) || sh.c < 8 || sh.c > kMaxC || sh.W < 3 || sh.W > 16 || sh.NB < 128) return H2_ERR_BATCH_SHAPE;
    const u32 m = (u32)scalars_n, cols = 2 * m, W = (u32)sh.W, NB = sh.NB, c = (u32)sh.c;
    int rc;
    static const bool fold_late = [] { const char *e = ab_env("H2_GG_FOLD_LATE"); return e && atoi(e) == 1; }();       // experiments (laboratory build)
    static const int lowprio = [] { const char *e = ab_env("H2_GG_LOWPRIO"); return e ? atoi(e) : 0; }();       // mask: 1 finish, 2 heavy, 4 rowcol, 8 planes (upper groups)
    static const int acc_block = [] { const char *e = ab_env("H2_GG_ACC_BLOCK"); return e && atoi(e) == 256 ? 256 : 512; }();       // 256: A/B
    static const int skip = [] { const char *e = ab_env("H2_GG_SKIP"); return e ? atoi(e) : 0; }();       // TIMING ONLY, WRONG RESULTS: drop side-stream fold stages (1 finish, 2 heavy, 4 rowcol, 8 planes, 16 chain)
    static const bool big_all = [] { const char *e = ab_env("H2_GG_BIG"); return !(e && atoi(e) == 0); }();
    static const int form_env = [] { const char *e = ab_env("H2_GG_FORM"); return e ? atoi(e) : 0; }();        // 1: latency form always, 2: throughput form always (A/B)
    // (from 2^22 points on the whole sort in front of a single group costs more than the grouping loses: 837 against 900 M scalar-mults/s on three streams)
    const bool throughput = form_env == 2 || (form_env != 1 && scalars_n < ((size_t)1 << 22) && msm_other_generic_in_flight(&cx));
    // ---- the groups, upper slices first.  Nine slices: 4 + 3 + 2 -- the first group's sort is all that stands in front of the first
    // addition, the last group's fold all that stands behind the last.  (H2_GENERIC_GROUPS="a,b,c": laboratory build only.)
    int sizes[MsmContext::kMaxGroups] = {0, 0, 0, 0}, G = 0;
    static const std::vector<int> env_sizes = [] {
        std::vector<int> v;
        if (const char *e = ab_env("H2_GENERIC_GROUPS"))
            for (const char *p = e; *p;) {
                v.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
        return v;
    }();
    {
        int sum = 0;
        for (int v : env_sizes) sum += v;
        if (!env_sizes.empty() && env_sizes.size() <= (size_t)MsmContext::kMaxGroups && sum == (int)W && *std::min_element(env_sizes.begin(), env_sizes.end()) >= 1) {
            for (int v : env_sizes) sizes[G++] = v;
        } else {
            if (throughput) {
                // independent calls on other streams are in flight: their sorts and folds hide beside this call's accumulate anyway, every accumulate
                // launch that starts beside them runs ragged, and side streams of several calls would share hardware queues: ONE group, everything on
                // the caller's stream (2^20: 1.16-1.22 ms per call on three streams against 1.26-1.46 for any grouping; 2^22: 862 M/s)
                sizes[0] = (int)W;
                G = 1;
            } else {
                // a call alone: 5 + 2 + 2 of nine slices -- the first group's sort is all that stands in front of the first addition, the last
                // (smallest) group's fold all that stands behind the last
                const int last = std::max(1, (int)W * 2 / 9), mid = last, top = (int)W - last - mid;
                sizes[0] = top; sizes[1] = mid; sizes[2] = last;
                G = 3;
            }
        }
    }
    Group grp[MsmContext::kMaxGroups];
    {
        u32 hi = W;
        size_t ent = 0, sts = 0, bkt = 0;
        for (int g = 0; g < G; ++g) {
            grp[g].ns = (u32)sizes[g];
            grp[g].w0 = hi - (u32)sizes[g];
            hi = grp[g].w0;
            if (!group_geometry(cols, NB, grp[g])) return H2_ERR_BATCH_SHAPE;
            grp[g].ent_off = ent;
            grp[g].starts_off = sts;
            grp[g].bucket_off = bkt;
            ent += grp[g].emax;
            sts += (size_t)grp[g].tb + 2;
            bkt += grp[g].tb;
        }
        if (ent >= ((size_t)1 << 31)) return H2_ERR_BATCH_SHAPE;
    }
    const u32 tb = W * NB, lane_div = 16;
    const double fraction = a.lane_fraction > 0.0 ? a.lane_fraction : g_lane_fraction.load();
    // The latency form leaves one CU per XCD without an accumulate workgroup and FENCES the chain kernels onto them: every accumulate workgroup asks
    // for 64 KiB of LDS it never touches, every link of the Horner chain for 100 KiB -- so a chain wave (one wave at raised priority issuing
    // dependent multiply-adds back to back: it takes ~90 % of its SIMD) can only land on a CU that holds no accumulate workgroup, instead of
    // starving two accumulate waves for its whole 130-180 us while every other lane of the launch waits for them (measured: the second group's
    // accumulate 494 -> 429 us; profiles/r06_generic_grouped.txt).  H2_GG_SPARE / H2_GG_LDS: laboratory build only.
    static const int spare_env = [] { const char *e = ab_env("H2_GG_SPARE"); return e ? atoi(e) : -1; }();
    static const int lds_env = [] { const char *e = ab_env("H2_GG_LDS"); return e ? atoi(e) : -1; }();
    const bool lds_fence = lds_env >= 0 ? lds_env == 1 : (!throughput && G > 1);
    const u32 spare = spare_env >= 0 ? (u32)spare_env : (lds_fence ? std::max(1u, lanes / 512u / 32u) : 0u);      // one CU in 32: 8 of an MI355X's 256 (one per XCD)
    const u32 usable = std::max(512u, (u32)(lanes * fraction) / 512u * 512u - 512u * std::min(spare, 64u));
    size_t head_slots = 0, hist_words = 0, tagged_words = 0, plan_words = 0;
    for (int g = 0; g < G; ++g) {
        grp[g].T = (u32)std::min<size_t>(usable, std::max<size_t>(512, (grp[g].emax / lane_div + 511) / 512 * 512));
        head_slots += grp[g].T;
        hist_words = std::max(hist_words, (size_t)grp[g].B1 * grp[g].gs.nh);
        tagged_words = std::max(tagged_words, grp[g].emax);
        plan_words = std::max(plan_words, grp[g].plan_words);
    }
    const int bb = (int)c - 1;
    const u32 wideS = 1u << (bb / 2), wideNR = NB / wideS;
    int cb = 0;
    while ((1u << cb) < wideS) ++cb;
    // ---- every workspace before anything is enqueued (a reservation that grows frees and synchronises)
    const u32 row = (cols + 7) & ~7u;
    if ((rc = cx.digits.reserve((size_t)W * row * 2 + 64)) != H2_OK) return rc;
    if ((rc = cx.hist.reserve(hist_words * 4)) != H2_OK) return rc;
    if ((rc = cx.tagged.reserve(tagged_words * 4)) != H2_OK) return rc;
    if ((rc = cx.plan.reserve(plan_words * 4)) != H2_OK) return rc;
    if ((rc = cx.starts.reserve(((size_t)tb + 2 * G) * 4)) != H2_OK) return rc;
    if ((rc = cx.entries.reserve((size_t)W * cols * 4)) != H2_OK) return rc;
    if ((rc = cx.heavy.reserve((size_t)G * (kMaxHeavy + 2) * 4)) != H2_OK) return rc;
    if ((rc = cx.hscratch.reserve((size_t)G * kMaxHeavy * kHeavyBlocks * 144)) != H2_OK) return rc;
    if ((rc = cx.partial.reserve((size_t)W * (wideS + wideNR + 32) * 144)) != H2_OK) return rc;
    if ((rc = cx.ssums.reserve((size_t)(W + G + 1) * 128)) != H2_OK) return rc;
    if ((rc = cx.seg9.reserve((head_slots + (size_t)tb) * 144)) != H2_OK) return rc;
    if ((rc = cx.bases9.reserve((size_t)m * 128 + 64)) != H2_OK) return rc;
    if (cx.fold_ctr.cap < (size_t)kMaxCols * 64) {      // fold9_planes' arrival counters: zero once, every launch leaves them at zero
        if ((rc = cx.fold_ctr.reserve((size_t)kMaxCols * 64)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.fold_ctr.ptr, 0, (size_t)kMaxCols * 64, st));
    }
    if (!cx.attr_grouped_set) {
        H2_HIP(hipFuncSetAttribute((const void *)msm_d1_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        H2_HIP(hipFuncSetAttribute((const void *)msm_s2_bins, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        H2_HIP(hipFuncSetAttribute((const void *)msm_combine<FB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        cx.attr_grouped_set = true;
    }
    // Hardware queues.  HIP multiplexes streams onto a handful of hardware queues (four per priority level), and every packet of a queue waits
    // for the one before it: a side stream that happens to share the CALLER's queue runs its kernels behind the accumulate they were meant to run
    // beside (measured inside bench.py, a dozen streams open: the second group's fold landed on the caller's queue, 1.85 ms per call against 1.55
    // from the native driver with the same library).  High-priority streams have queues of their own but cost every other stream of the process
    // (independent calls 1.19 -> 1.32 ms with one such stream merely open: profiles/r06_generic_grouped.txt).  So the latency form runs on THREE
    // streams of its own, created back to back (the runtime hands a new stream the least-used queue: three in a row get different ones): one
    // carries the accumulates and what is serial with them, two side streams the rest (the conversion, the later groups' sorts and the even groups'
    // folds and chain links on one, the odd groups' on the other: a group's fold must not queue behind the previous group's chain link); the caller's stream waits for `ms` at the end and does
    // nothing in between, so it does not matter whose queue it shares.  The throughput form (one group) has nothing beside its accumulate and
    // stays on the caller's stream.
    if (G > 1 && !cx.gstream[0])
        for (int i = 0; i < 3; ++i) H2_HIP(hipStreamCreateWithFlags(&cx.gstream[i], hipStreamNonBlocking));
    hipStream_t const caller = st;
    if (G > 1) st = cx.gstream[1];
    for (int i = 0; i < 4 + 3 * G; ++i)
        if (!cx.gev[i]) H2_HIP(hipEventCreateWithFlags(&cx.gev[i], hipEventDisableTiming));
    hipStream_t sort_s = cx.gstream[0];
    hipEvent_t ev_fork = cx.gev[0], ev_conv = cx.gev[1];
    auto ev_sorted = [&](int g) { return cx.gev[2 + 3 * g]; };
    auto ev_acc = [&](int g) { return cx.gev[3 + 3 * g]; };
    auto ev_chain = [&](int g) { return cx.gev[4 + 3 * g]; };
    auto fold_s = [&](int g) { return g == G - 1 ? st : cx.gstream[(g & 1) ? 2 : 0]; };          // the last group folds behind its accumulate

    ColIn ci;
    ColOut co;
    ColStride cs;
    memset(&ci, 0, sizeof ci);
    memset(&co, 0, sizeof co);
    memset(&cs, 0, sizeof cs);
    const ColStride cs0 = cs;
    const bool mont = a.form == H2_FORM_MONTGOMERY;
    uint16_t *digits = cx.digits.as<uint16_t>();
    u32 *hist1 = cx.hist.as<u32>(), *tagged = cx.tagged.as<u32>();
    u32 *heads_all = cx.seg9.as<u32>(), *buckets_all = heads_all + 36 * head_slots;
    u32 *lines9 = cx.partial.as<u32>(), *planes9 = lines9 + 36 * (size_t)W * (wideS + wideNR), *ssums = cx.ssums.as<u32>();
    const u32 *pts = cx.bases9.as<u32>();

    if (st != caller) {               // everything below waits for what the caller's stream holds now
        H2_HIP(hipEventRecord(cx.gev[2 + 3 * G], caller));
        H2_HIP(hipStreamWaitEvent(st, cx.gev[2 + 3 * G], 0));
    }
    // ---- the bases' conversion (it reads nothing the sorts write) beside the recode and the first sort
    H2_HIP(hipEventRecord(ev_fork, st));
    H2_HIP(hipStreamWaitEvent(fold_s(0), ev_fork, 0));
    hipLaunchKernelGGL((msm_bases_to_m9_glv<FB>), dim3((m + 255) / 256), dim3(256), 0, fold_s(0), (const u32 *)a.d_bases, cx.bases9.as<u32>(), m);
    H2_HIP(hipEventRecord(ev_conv, fold_s(0)));
    // ---- the endomorphism split, once
    hipLaunchKernelGGL((msm_glv_digits<FS>), dim3((m + 255) / 256), dim3(256), 0, st, (const u32 *)a.d_scalars, digits, m, row, (int)c, (int)W, mont ? 1 : 0);

    size_t heads_off = 0;
    u32 *heads_of[MsmContext::kMaxGroups];
    for (int g = 0; g < G; ++g) {
        heads_of[g] = heads_all + 36 * heads_off;
        heads_off += grp[g].T;
    }
    auto sort_group = [&](int g, hipStream_t s_) {
        const Group &q = grp[g];
        u32 *bin_count = cx.plan.as<u32>(), *bin_start = bin_count + q.gs.nh, *big = bin_start + q.gs.nh + 1, *gcnt = big + 64;
        u32 *starts = cx.starts.as<u32>() + q.starts_off, *entries = cx.entries.as<u32>() + q.ent_off, *heavy = cx.heavy.as<u32>() + (size_t)g * (kMaxHeavy + 2);
        const size_t nbk = (size_t)1 << q.gs.lowb;
        hipLaunchKernelGGL(msm_d1_count, dim3(q.B1), dim3(512), q.gs.nh * 4, s_, (const uint16_t *)digits, q.gs, hist1);
        hipLaunchKernelGGL(msm_s1_prefix, dim3((q.gs.nh + 15) / 16), dim3(1024), 0, s_, hist1, bin_count, q.B1, q.gs.nh, heavy, big, starts + q.tb + 1, cs);
        hipLaunchKernelGGL(msm_d1_scatter, dim3(q.B1), dim3(512), ((size_t)q.gs.nh * 3 + 1 + (size_t)q.gs.S * q.ns) * 4, s_, (const uint16_t *)digits, q.gs,
                           (const u32 *)hist1, (const u32 *)bin_count, bin_start, tagged);
        // pass 2: a workgroup per bin; it also clears the group's raw bucket slots.  A bin beyond its stage (the carry slice of the split
        // is ONE bucket; columns of repeated scalars) goes to the chunked msm_s2_big_* kernels, which return at once when the list is empty
        hipLaunchKernelGGL(msm_s2_bins, dim3(q.gs.nh), dim3(1024), (nbk * 2 + q.cap) * 4, s_, (const u32 *)tagged, (const uint16_t *)nullptr, (const u32 *)bin_start, q.p2,
                           q.tb, (u32)q.cap, starts, entries, big, (big_all || !g) ? kMaxBig : 0u, buckets_all + 36 * q.bucket_off, cs);
        if (!big_all && g) return;
        hipLaunchKernelGGL(msm_s2_big_count, dim3(kBigChunks, kMaxBig), dim3(256), nbk * 4, s_, (const u32 *)tagged, (const uint16_t *)nullptr, (const u32 *)bin_start, q.p2,
                           (const u32 *)big, gcnt, cs);
        hipLaunchKernelGGL(msm_s2_big_prefix, dim3(kMaxBig), dim3(256), nbk * 4, s_, (const u32 *)bin_start, q.p2, q.tb, (const u32 *)big, gcnt, starts, cs);
        hipLaunchKernelGGL(msm_s2_big_scatter, dim3(kBigChunks, kMaxBig), dim3(256), nbk * 4, s_, (const u32 *)tagged, (const uint16_t *)nullptr, (const u32 *)bin_start, q.p2,
                           (const u32 *)big, (const u32 *)gcnt, entries, cs);
    };
    // a group's fold down to its slice sums: finish (range heads into their buckets), the heavy buckets, line sums, planes
    auto fold_group = [&](int g, hipStream_t s_) {
        const Group &q = grp[g];
        const u32 *starts = cx.starts.as<u32>() + q.starts_off;
        u32 *gbuckets = buckets_all + 36 * q.bucket_off, *heavy = cx.heavy.as<u32>() + (size_t)g * (kMaxHeavy + 2);
        u32 *hscr = cx.hscratch.as<u32>() + (size_t)g * kMaxHeavy * kHeavyBlocks * 36;
        const int sk = g < G - 1 ? skip : 0, lp = g < G - 1 ? lowprio : 0;
        ColStride cs = cs0, csf = cs0, csh = cs0, csr = cs0, csp = cs0;
        csf.lowprio = (lp & 1) ? 1u : 0u;
        csh.lowprio = (lp & 2) ? 1u : 0u;
        csr.lowprio = (lp & 4) ? 1u : 0u;
        csp.lowprio = (lp & 8) ? 1u : 0u;
        (void)cs;
        if (!(sk & 1)) hipLaunchKernelGGL((fold9_finish<FB>), dim3((q.tb + 255) / 256), dim3(256), 0, s_, (const u32 *)heads_of[g], starts, gbuckets, heavy, q.tb, q.T, lane_div, csf);
        if (!(sk & 2)) hipLaunchKernelGGL((fold9_finish_heavy<FB>), dim3(kHeavyBlocks, kHeavyRows), dim3(256), 0, s_, (const u32 *)heads_of[g], starts, hscr, (const u32 *)heavy, q.tb, q.T, lane_div, csh);
        if (!(sk & 2)) hipLaunchKernelGGL((fold9_finish_heavy2<FB>), dim3(kHeavyRows), dim3(64), 0, s_, (const u32 *)hscr, gbuckets, (const u32 *)heavy, csh);
        if (!(sk & 4)) hipLaunchKernelGGL((fold9_rowcol<FB>), dim3(wideS + wideNR - 1, q.ns), dim3(256), 0, s_, (const u32 *)gbuckets, lines9 + 36 * (size_t)q.w0 * (wideS + wideNR), wideS, wideNR, csr);
        if (!(sk & 8)) hipLaunchKernelGGL((fold9_planes<FB>), dim3(c - 1, q.ns), dim3(256), 0, s_, (const u32 *)(lines9 + 36 * (size_t)q.w0 * (wideS + wideNR)), planes9 + 36 * (size_t)q.w0 * 32,
                           cx.fold_ctr.as<u32>() + q.w0, wideS, wideNR, cb, ssums + 32 * (size_t)q.w0, kOutSliceSum, mont, co, csp);
    };

    // ---- sorts: the first group on the caller's stream, the others on the side stream behind it (they share hist / tagged / plan)
    sort_group(0, st);
    H2_HIP(hipEventRecord(ev_sorted(0), st));
    H2_HIP(hipStreamWaitEvent(sort_s, ev_sorted(0), 0));
    for (int g = 1; g < G; ++g) {
        sort_group(g, sort_s);
        H2_HIP(hipEventRecord(ev_sorted(g), sort_s));
    }
    // ---- accumulates back to back on the caller's stream; every group's fold and its link of the chain beside the next accumulate
    H2_HIP(hipStreamWaitEvent(st, ev_conv, 0));
    for (int g = 0; g < G; ++g) {
        const Group &q = grp[g];
        if (g) H2_HIP(hipStreamWaitEvent(st, ev_sorted(g), 0));
        if (acc_block == 512)
            hipLaunchKernelGGL((msm_accumulate<FB, false, true, 512>), dim3(q.T / 512), dim3(512), lds_fence ? 65536 : 0, st, pts, (const u32 *)nullptr, 0xFFFFFFFFu,
                               (const u32 *)(cx.entries.as<u32>() + q.ent_off), (const u32 *)(cx.starts.as<u32>() + q.starts_off), heads_of[g],
                               buckets_all + 36 * q.bucket_off, q.tb, q.T, lane_div, cs);
        else
        hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(q.T / 256), dim3(256), 0, st, pts, (const u32 *)nullptr, 0xFFFFFFFFu,
                           (const u32 *)(cx.entries.as<u32>() + q.ent_off), (const u32 *)(cx.starts.as<u32>() + q.starts_off), heads_of[g],
                           buckets_all + 36 * q.bucket_off, q.tb, q.T, lane_div, cs);
        if (g < G - 1 && !fold_late) {
            hipStream_t fs = fold_s(g);
            H2_HIP(hipEventRecord(ev_acc(g), st));
            H2_HIP(hipStreamWaitEvent(fs, ev_acc(g), 0));
            fold_group(g, fs);
            if (g) H2_HIP(hipStreamWaitEvent(fs, ev_chain(g - 1), 0));
            // A_g = Horner(group g) + D_(g-1);  D_g = 2^(c ns_(g+1)) A_g, an XYZZ point behind the slice sums
            if (!(skip & 16)) hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), lds_fence ? 100 * 1024 : 0, fs, (const u32 *)(ssums + 32 * (size_t)q.w0), (int)q.ns, (int)c, ssums + 32 * (size_t)(W + g), kOutSliceSum, 1,
                               (int)(c * grp[g + 1].ns), g ? (const u32 *)(ssums + 32 * (size_t)(W + g - 1)) : (const u32 *)nullptr, 1);
            H2_HIP(hipEventRecord(ev_chain(g), fs));
        }
    }
    if (fold_late)
        for (int g = 0; g < G - 1; ++g) {
            fold_group(g, st);
            hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, (const u32 *)(ssums + 32 * (size_t)grp[g].w0), (int)grp[g].ns, (int)c, ssums + 32 * (size_t)(W + g), kOutSliceSum, 1,
                               (int)(c * grp[g + 1].ns), g ? (const u32 *)(ssums + 32 * (size_t)(W + g - 1)) : (const u32 *)nullptr, 1);
        }
    fold_group(G - 1, st);
    if (G > 1 && !fold_late) H2_HIP(hipStreamWaitEvent(st, ev_chain(G - 2), 0));
    hipLaunchKernelGGL((msm_combine<FB>), dim3(1), dim3(64), 0, st, (const u32 *)(ssums + 32 * (size_t)grp[G - 1].w0), (int)grp[G - 1].ns, (int)c, (u32 *)a.d_out, a.out_kind,
                       mont ? 1 : 0, 0, G > 1 ? (const u32 *)(ssums + 32 * (size_t)(W + G - 2)) : (const u32 *)nullptr, 1);
    H2_HIP(hipGetLastError());
    if (st != caller) {               // ... and the caller's stream for the result
        H2_HIP(hipEventRecord(cx.gev[3 + 3 * G], st));
        H2_HIP(hipStreamWaitEvent(caller, cx.gev[3 + 3 * G], 0));
        st = caller;
    }
    if (!cx.gdone) H2_HIP(hipEventCreateWithFlags(&cx.gdone, hipEventDisableTiming));
    H2_HIP(hipEventRecord(cx.gdone, st));
    cx.gpending = true;
    return H2_OK;
}

template int msm_generic_grouped<FP, FQ>(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st);
template int msm_generic_grouped<FQ, FP>(MsmContext &cx, const MsmArgs &a, const MsmShape &sh, size_t scalars_n, u32 lanes, hipStream_t st);

}  // namespace h2
