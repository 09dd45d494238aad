// Multiexps and commits whose inputs are HOST slices (what a Rust shim passes): PCIe pipelines in front of the device path of msm_launch.hip.
#include "msm_internal.cuh"

using namespace h2;

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
    static const int env = [] { const char *e = ab_env("H2_MSM_HOST_CHUNKS"); return e ? atoi(e) : 0; }();
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
    // Range boundaries.  Equal ranges by default; H2_MSM_HOST_SPLIT="a,b,c" (laboratory build; Q percentages) makes them uneven -- a smaller LAST
    // range shortens what stands behind the last byte (its accumulate), as long as the ranges before it keep pace with their copies.
    static const std::vector<int> split_env = [] {
        std::vector<int> v;
        if (const char *e = ab_env("H2_MSM_HOST_SPLIT"))
            for (const char *q = e; *q;) {
                v.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        return v;
    }();
    size_t bnd[17];
    {
        int tot = 0;
        for (int v : split_env) tot += v;
        const bool uneven = split_env.size() == Q && tot == 100 && *std::min_element(split_env.begin(), split_env.end()) >= 1;
        size_t acc_pct = 0;
        bnd[0] = 0;
        for (unsigned q = 0; q < Q; ++q) {
            if (uneven) { acc_pct += (size_t)split_env[q]; bnd[q + 1] = q + 1 == Q ? n : (n * acc_pct / 100) & ~(size_t)63; }
            else bnd[q + 1] = n * (q + 1) / Q;
        }
    }
    auto range = [&](unsigned q, size_t &lo, size_t &hi) { lo = bnd[q]; hi = bnd[q + 1]; };
    size_t widest = 0;
    for (unsigned q = 0; q < Q; ++q) widest = std::max(widest, bnd[q + 1] - bnd[q]);
    const int c = choose_c(widest, false);                     // ONE window width for every range: their slice sums add up
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
    static const bool graphs_on = [] { const char *e = ab_env("H2_MSM_HOST_GRAPHS"); return !(e && atoi(e) == 0); }();
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
    static const bool thread_on = [] { const char *e = ab_env("H2_MSM_HOST_THREAD"); return e && atoi(e) == 1; }();
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
    static const bool overlap_on = [] { const char *e = ab_env("H2_MSM_HOST_OVERLAP"); return !(e && atoi(e) == 0); }();
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
