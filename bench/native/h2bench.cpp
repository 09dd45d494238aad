// h2bench: native (no Python) timing + parity driver of the hot path through the C ABI (include/halo2_mi355x.h).
//
// Why: a Python-driven run on a fresh GPU box pays 1-2 minutes of `import torch` before the first kernel; this binary loads the
// library, the C oracle (the checker: oracle/h2_oracle.c, test infrastructure) and the HIP runtime and is measuring within a second,
// so a same-box A/B of two library builds costs seconds of GPU time:
//
//     build/h2bench commit [log_n=20] [steps=20] [warmup=5] [streams=3] [curve=0|1] [reps=1]   the bench line's workload: registered table, blinds,
//                                                                               independent column commits round-robin over streams; `steps` and
//                                                                               `streams` may be lists (20,100 and 2,3,4): every pair, median of reps
//     build/h2bench ntt [sizes=16,18,20,22] [field=0|1] [check=1]                    device-resident best_fft: warm timing, elementwise parity
//     build/h2bench msm [log_n=20] [curve=0|1]                                       generic best_multiexp (no registered table): time + parity
//     build/h2bench host [log_n=20]                                                   the host-pointer seam: h2_msm / h2_ntt / h2_commit incl. PCIe
//     build/h2bench batch [log_n=13] [columns=8] [curve=0|1]                          h2_commit_batch_device: K columns + blinds in one call, against K single commits
//     build/h2bench domain [k=20] [ext=1|2] [field=0|1]                               ifft / coeff_to_extended / extended_to_coeff, device-resident: parity + time
//     build/h2bench parity                                                           a sweep of small and odd sizes through every entry point above
//
// Every mode checks its results against the oracle (bit-exact: canonical affine coordinates / every element) and prints one summary
// line per measurement plus a final `H2BENCH OK` or `H2BENCH FAIL`.  H2BENCH_LIB=<path> loads another build of the library (A/B).
//
// Build (see __graft_entry__.build): hipcc -O2 -std=c++17 bench/native/h2bench.cpp oracle/h2_oracle.c -ldl -lpthread -o build/h2bench
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/halo2_mi355x.h"

extern "C" {
void orc_random_field(int field, uint64_t seed, uint64_t *out, size_t n);
void orc_generate_bases(int curve, const uint64_t *g_xy, uint64_t seed, uint64_t *out_xy, size_t n);
int orc_best_multiexp(int curve, const uint64_t *scalars, const uint64_t *bases, size_t n, uint64_t *out_xyz);
int orc_commit(int curve, const uint64_t *g, const uint64_t *w, const uint64_t *poly, const uint64_t *blind, size_t n, uint64_t *out_xyz);
void orc_point_to_affine(int curve, uint64_t *out_xy, const uint64_t *in_xyz);
int orc_best_fft(int field, uint64_t *a, const uint64_t *omega, unsigned log_n);
void orc_to_mont(int field, uint64_t *a, size_t n);
int orc_ifft(int field, uint64_t *a, const uint64_t *omega_inv, unsigned log_n, const uint64_t *divisor);
int orc_coeff_to_extended(int field, uint64_t *a_ext, unsigned k, unsigned ext_k, const uint64_t *g_coset, const uint64_t *g_coset_inv, const uint64_t *extended_omega);
int orc_extended_to_coeff(int field, uint64_t *a_ext, unsigned ext_k, const uint64_t *g_coset, const uint64_t *g_coset_inv, const uint64_t *extended_omega_inv,
                          const uint64_t *extended_ifft_divisor);
}

// ---- the library, bound at run time so that H2BENCH_LIB can point at another build --------------------------------------------
#define H2_FN(name) static decltype(&::name) p_##name
H2_FN(h2_init); H2_FN(h2_last_error); H2_FN(h2_device_count); H2_FN(h2_bases_register_ex); H2_FN(h2_commit_column_window_bits);
H2_FN(h2_bases_set_blind_base); H2_FN(h2_bases_free); H2_FN(h2_commit_device); H2_FN(h2_commit); H2_FN(h2_msm_device); H2_FN(h2_msm);
H2_FN(h2_ntt_device); H2_FN(h2_ntt); H2_FN(h2_profile_enable); H2_FN(h2_profile_read); H2_FN(h2_profile_read_busy); H2_FN(h2_commit_batch_device);
H2_FN(h2_ifft_device); H2_FN(h2_coeff_to_extended_device); H2_FN(h2_extended_to_coeff_device);
static bool load_library(const char *argv0) {
    std::string path;
    if (const char *e = getenv("H2BENCH_LIB")) path = e;
    else {
        std::string self = argv0;
        const size_t cut = self.rfind('/');
        path = (cut == std::string::npos ? std::string(".") : self.substr(0, cut)) + "/../halo2_amd/libhalo2_mi355x.so";
    }
    void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "h2bench: cannot load %s: %s\n", path.c_str(), dlerror()); return false; }
#define H2_BIND(name) if (!(p_##name = (decltype(p_##name))dlsym(lib, #name))) { fprintf(stderr, "h2bench: %s lacks %s\n", path.c_str(), #name); return false; }
    H2_BIND(h2_init) H2_BIND(h2_last_error) H2_BIND(h2_device_count) H2_BIND(h2_bases_register_ex) H2_BIND(h2_commit_column_window_bits)
    H2_BIND(h2_bases_set_blind_base) H2_BIND(h2_bases_free) H2_BIND(h2_commit_device) H2_BIND(h2_commit) H2_BIND(h2_msm_device) H2_BIND(h2_msm)
    H2_BIND(h2_ntt_device) H2_BIND(h2_ntt) H2_BIND(h2_profile_enable) H2_BIND(h2_profile_read) H2_BIND(h2_profile_read_busy) H2_BIND(h2_commit_batch_device)
    H2_BIND(h2_ifft_device) H2_BIND(h2_coeff_to_extended_device) H2_BIND(h2_extended_to_coeff_device)
    printf("library: %s\n", path.c_str());
    return true;
}

static int g_fail = 0;
#define CHECK_RC(call)                                                                         \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != H2_OK) {                                                                    \
            printf("FAIL: %s -> %d (%s)\n", #call, rc_, p_h2_last_error());                    \
            g_fail++;                                                                          \
            return;                                                                            \
        }                                                                                      \
    } while (0)
#define HIPCK(call)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            printf("FAIL: %s -> %s\n", #call, hipGetErrorString(e_));                          \
            g_fail++;                                                                          \
            return;                                                                            \
        }                                                                                      \
    } while (0)
static void expect(bool ok, const char *what) {
    if (!ok) { printf("FAIL: %s\n", what); g_fail++; }
    else printf("ok: %s\n", what);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// generator the bases are seeded multiples of: (-1, 2), the on-curve point pinned at poly/commitment/msm.rs:181
static void generator(int curve, uint64_t gen[8]) {
    static const uint64_t P_FP[4] = {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL};
    static const uint64_t P_FQ[4] = {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL};
    const int base_field = curve == H2_PALLAS ? H2_FP : H2_FQ;       // coordinates of Pallas live in Fp, of Vesta in Fq
    const uint64_t *p = base_field == H2_FP ? P_FP : P_FQ;
    memset(gen, 0, 64);
    gen[0] = p[0] - 1; gen[1] = p[1]; gen[2] = p[2]; gen[3] = p[3];
    gen[4] = 2;
    orc_to_mont(base_field, gen, 2);
}
static int scalar_field(int curve) { return curve == H2_PALLAS ? H2_FQ : H2_FP; }
static bool same_point(int curve, const uint64_t *a_xyz, const uint64_t *b_xyz) {
    uint64_t x[8], y[8];
    orc_point_to_affine(curve, x, a_xyz);
    orc_point_to_affine(curve, y, b_xyz);
    return memcmp(x, y, 64) == 0;
}

// ---- shader clock and socket power while a region runs (amdgpu hwmon of the first GPU that has one: freq1_input in Hz, power1_input
// in uW; sampled every ~2 ms from a thread -- what bench.py's `clock` object reads).  The sustained commit rate is a clock story.
#include <glob.h>
#include <atomic>
#include <thread>
struct ClockWatch {
    std::string dir;
    std::vector<double> f, p;
    std::atomic<bool> stop{false};
    std::thread th;
    ClockWatch() {
        // the hwmon node of the GPU HIP device 0 is (sysfs shows every card of the host, the process sees one): by PCI address
        char bus[32] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return;
        for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
        const std::string pat = std::string("/sys/bus/pci/devices/") + bus + "/hwmon/hwmon*/freq1_input";
        glob_t g;
        if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc) {
            dir = g.gl_pathv[0];
            dir.resize(dir.rfind('/'));
        }
        globfree(&g);
    }
    static double read1(const std::string &path) {
        FILE *fh = fopen(path.c_str(), "r");
        double v = 0;
        if (fh) { if (fscanf(fh, "%lf", &v) != 1) v = 0; fclose(fh); }
        return v;
    }
    void start() {
        if (dir.empty()) return;
        f.clear(); p.clear(); stop = false;
        th = std::thread([this] {
            while (!stop) {
                f.push_back(read1(dir + "/freq1_input") / 1e6);
                p.push_back(read1(dir + "/power1_input") / 1e6);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        });
    }
    std::string finish() {
        if (dir.empty()) return "clock: no hwmon";
        stop = true;
        th.join();
        if (f.empty()) return "clock: no samples";
        std::sort(f.begin(), f.end()); std::sort(p.begin(), p.end());
        char buf[160];
        snprintf(buf, sizeof buf, "sclk %.0f MHz (%.0f..%.0f), %.0f W (%zu samples)", f[f.size() / 2], f.front(), f.back(), p[p.size() / 2], f.size());
        return buf;
    }
};

// ---- commit: the bench line's workload ------------------------------------------------------------------------------------------
static void mode_commit(unsigned log_n, const std::vector<unsigned> &steps_list, int warmup, const std::vector<unsigned> &stream_list, int curve, int reps = 1) {
    const size_t n = (size_t)1 << log_n;
    const int sf = scalar_field(curve), ncols = 4;
    uint64_t gen[8];
    generator(curve, gen);
    std::vector<uint64_t> bases(n * 8), w(8), blinds((size_t)ncols * 4);
    std::vector<std::vector<uint64_t>> cols(ncols, std::vector<uint64_t>(n * 4));
    double t0 = now_ms();
    orc_generate_bases(curve, gen, 0x48414C4F32, bases.data(), n);
    orc_generate_bases(curve, gen, 0x77, w.data(), 1);
    for (int c = 0; c < ncols; ++c) orc_random_field(sf, 1000 + c, cols[c].data(), n);
    orc_random_field(sf, 0xB11D, blinds.data(), ncols);
    printf("inputs: 2^%u points, %d columns generated in %.2f s\n", log_n, ncols, (now_ms() - t0) / 1e3);
    h2_bases_t g = 0;
    const int c_bits = p_h2_commit_column_window_bits(n);
    t0 = now_ms();
    CHECK_RC(p_h2_bases_register_ex(curve, bases.data(), n, H2_FORM_MONTGOMERY, c_bits, &g));
    HIPCK(hipDeviceSynchronize());
    printf("h2_bases_register_ex (%d-bit windows): %.1f ms\n", c_bits, now_ms() - t0);
    CHECK_RC(p_h2_bases_set_blind_base(g, w.data(), H2_FORM_MONTGOMERY));
    std::vector<void *> d_cols(ncols);
    void *d_blinds = nullptr, *d_out = nullptr;
    for (int c = 0; c < ncols; ++c) {
        HIPCK(hipMalloc(&d_cols[c], n * 32));
        HIPCK(hipMemcpy(d_cols[c], cols[c].data(), n * 32, hipMemcpyHostToDevice));
    }
    HIPCK(hipMalloc(&d_blinds, (size_t)ncols * 32));
    HIPCK(hipMemcpy(d_blinds, blinds.data(), (size_t)ncols * 32, hipMemcpyHostToDevice));
    int max_steps = 8;
    for (unsigned k : steps_list) max_steps = std::max<int>(max_steps, (int)k);
    const int outs = max_steps;
    HIPCK(hipMalloc(&d_out, (size_t)outs * 96));
    unsigned max_streams = 1;
    for (unsigned k : stream_list) max_streams = std::max(max_streams, k);
    std::vector<hipStream_t> st(max_streams);
    for (auto &s : st) HIPCK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    bool launch_failed = false;
    int nstreams = (int)stream_list[0];
    auto step = [&](int i) {
        const int c = i % ncols;
        const int rc = p_h2_commit_device(g, d_cols[c], n, nullptr, (char *)d_blinds + 32 * c, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN,
                                          (char *)d_out + 96 * (size_t)(i % outs), st[i % nstreams]);
        if (rc != H2_OK && !launch_failed) { printf("FAIL: h2_commit_device -> %d (%s)\n", rc, p_h2_last_error()); launch_failed = true; g_fail++; }
    };
    // every (streams, steps) pair of the lists, `reps` timed regions each (the median is printed last); before each region the clocks
    // settle under the same schedule (400 ms before the first, 150 ms before the others: bench.py --prewarm-ms) and `warmup` steps run untimed
    bool first_region = true;
    for (unsigned S : stream_list) {
        nstreams = (int)S;
        for (int i = 0; i < 2 * nstreams; ++i) step(i);           // workspaces are allocated on first use
        HIPCK(hipDeviceSynchronize());
        for (unsigned steps : steps_list) {
            std::vector<double> per_step;
            double union_ms = 0, mean_ms = 0;
            std::string clock_note;
            for (int rep = 0; rep < reps; ++rep) {
                t0 = now_ms();
                while (now_ms() - t0 < (first_region ? 400.0 : 150.0)) {
                    for (int i = 0; i < 2 * nstreams; ++i) step(i);
                    HIPCK(hipDeviceSynchronize());
                }
                first_region = false;
                for (int i = 0; i < warmup; ++i) step(i);
                HIPCK(hipDeviceSynchronize());
                p_h2_profile_enable(2);
                ClockWatch cw;
                if (rep == reps - 1 && steps >= 200) cw.start();            // long regions only: a 20 ms region is ten samples
                t0 = now_ms();
                for (unsigned i = 0; i < steps; ++i) step((int)i);
                HIPCK(hipDeviceSynchronize());
                const double ms = now_ms() - t0;
                if (rep == reps - 1 && steps >= 200) clock_note = cw.finish();
                double tot = 0, busy = 0;
                uint64_t cnt = 0;
                p_h2_profile_read_busy(H2_PROF_MSM_ACCUMULATE, &tot, &busy, &cnt);
                p_h2_profile_enable(0);
                per_step.push_back(ms / steps);
                union_ms = cnt ? busy / cnt : 0.0;
                mean_ms = cnt ? tot / cnt : 0.0;
            }
            std::sort(per_step.begin(), per_step.end());
            const double med = per_step[per_step.size() / 2];
            printf("commit 2^%u x %u steps over %d streams: %.4f ms per step = %.1f M scalar-mults/s   (median of %d regions, %.4f .. %.4f; accumulate of the last: union %.4f ms per launch, mean %.4f)\n",
                   log_n, steps, nstreams, med, n / med / 1e3, reps, per_step.front(), per_step.back(), union_ms, mean_ms);
            if (!clock_note.empty()) printf("    under that load: %s\n", clock_note.c_str());
        }
    }
    // one commit at a time on one stream
    for (int i = 0; i < 4; ++i) { const int rc = p_h2_commit_device(g, d_cols[0], n, nullptr, d_blinds, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_out, st[0]); (void)rc; }
    HIPCK(hipDeviceSynchronize());
    p_h2_profile_enable(1);
    t0 = now_ms();
    const int lone = 20;
    for (int i = 0; i < lone; ++i) {
        const int c = i % ncols;
        CHECK_RC(p_h2_commit_device(g, d_cols[c], n, nullptr, (char *)d_blinds + 32 * c, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_out, st[0]));
    }
    HIPCK(hipDeviceSynchronize());
    const double lone_ms = (now_ms() - t0) / lone;
    double stage[4] = {0, 0, 0, 0};
    for (int slot : {H2_PROF_MSM_ACCUMULATE, H2_PROF_MSM_SORT, H2_PROF_MSM_REDUCE}) {
        double t = 0;
        uint64_t k = 0;
        p_h2_profile_read(slot, &t, &k);
        stage[slot] = k ? t / k : 0;
    }
    p_h2_profile_enable(0);
    printf("one commit at a time: %.4f ms   (sort %.4f + accumulate %.4f + fold %.4f)\n", lone_ms, stage[H2_PROF_MSM_SORT], stage[H2_PROF_MSM_ACCUMULATE],
           stage[H2_PROF_MSM_REDUCE]);
    // parity of column 1 (the last lone commit used column (lone - 1) % ncols) against Params::commit of the oracle
    {
        const int c = 1;
        CHECK_RC(p_h2_commit_device(g, d_cols[c], n, nullptr, (char *)d_blinds + 32 * c, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_out, st[0]));
        HIPCK(hipDeviceSynchronize());
        uint64_t got[12], want[12];
        HIPCK(hipMemcpy(got, d_out, 96, hipMemcpyDeviceToHost));
        orc_commit(curve, bases.data(), w.data(), cols[c].data(), blinds.data() + 4 * c, n, want);
        expect(same_point(curve, got, want), "h2_commit_device == oracle Params::commit (with blind)");
    }
    for (auto &s : st) (void)hipStreamDestroy(s);
    for (void *p : d_cols) (void)hipFree(p);
    (void)hipFree(d_blinds);
    (void)hipFree(d_out);
    p_h2_bases_free(g);
}

// ---- ntt ----------------------------------------------------------------------------------------------------------------------
static void mode_ntt(const std::vector<unsigned> &sizes, int field, bool check) {
    for (unsigned L : sizes) {
        const size_t n = (size_t)1 << L;
        std::vector<uint64_t> a(n * 4), omega(4);
        orc_random_field(field, 3 + L, a.data(), n);
        orc_random_field(field, 9000 + L, omega.data(), 1);       // any omega: the network is the reference's (benches/fft.rs:17 uses a random one)
        void *d = nullptr;
        HIPCK(hipMalloc(&d, n * 32));
        HIPCK(hipMemcpy(d, a.data(), n * 32, hipMemcpyHostToDevice));
        if (check) {
            CHECK_RC(p_h2_ntt_device(field, d, L, omega.data(), H2_FORM_MONTGOMERY, nullptr));
            HIPCK(hipDeviceSynchronize());
            std::vector<uint64_t> got(n * 4), want = a;
            HIPCK(hipMemcpy(got.data(), d, n * 32, hipMemcpyDeviceToHost));
            orc_best_fft(field, want.data(), omega.data(), L);
            char msg[96];
            snprintf(msg, sizeof msg, "h2_ntt_device 2^%u (field %d) == oracle best_fft at every index", L, field);
            expect(got == want, msg);
        }
        for (int i = 0; i < 30; ++i) CHECK_RC(p_h2_ntt_device(field, d, L, omega.data(), H2_FORM_MONTGOMERY, nullptr));
        HIPCK(hipDeviceSynchronize());
        double best = 1e30;
        const int R = 40;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now_ms();
            for (int i = 0; i < R; ++i) CHECK_RC(p_h2_ntt_device(field, d, L, omega.data(), H2_FORM_MONTGOMERY, nullptr));
            HIPCK(hipDeviceSynchronize());
            best = std::min(best, (now_ms() - t0) / R);
        }
        const double bf = (double)(n / 2) * L;
        printf("ntt 2^%u: %.4f ms  %.1f G butterflies/s\n", L, best, bf / best / 1e6);
        if (getenv("H2BENCH_CLOCK") && L >= 18) {      // shader clock / socket power with transforms back to back for ~0.4 s
            ClockWatch cw;
            cw.start();
            const double tc = now_ms();
            int done = 0;
            while (now_ms() - tc < 400.0) {
                for (int i = 0; i < 50; ++i) CHECK_RC(p_h2_ntt_device(field, d, L, omega.data(), H2_FORM_MONTGOMERY, nullptr));
                HIPCK(hipDeviceSynchronize());
                done += 50;
            }
            const double per = (now_ms() - tc) / done;
            printf("    back to back for 0.4 s: %.4f ms per transform; %s\n", per, cw.finish().c_str());
        }
        (void)hipFree(d);
    }
}

// ---- generic best_multiexp ----------------------------------------------------------------------------------------------------
static void mode_msm_n(size_t n, int curve, bool timing) {
    const int sf = scalar_field(curve);
    uint64_t gen[8];
    generator(curve, gen);
    std::vector<uint64_t> bases(std::max<size_t>(n, 1) * 8), sc(std::max<size_t>(n, 1) * 4);
    orc_generate_bases(curve, gen, 0x4D534D + n, bases.data(), n);
    orc_random_field(sf, 77 + n, sc.data(), n);
    void *d_b = nullptr, *d_s = nullptr, *d_o = nullptr;
    HIPCK(hipMalloc(&d_b, std::max<size_t>(n, 1) * 64));
    HIPCK(hipMalloc(&d_s, std::max<size_t>(n, 1) * 32));
    HIPCK(hipMalloc(&d_o, 96));
    HIPCK(hipMemcpy(d_b, bases.data(), n * 64, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_s, sc.data(), n * 32, hipMemcpyHostToDevice));
    CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_o, nullptr));
    HIPCK(hipDeviceSynchronize());
    uint64_t got[12], want[12], got_h[12];
    HIPCK(hipMemcpy(got, d_o, 96, hipMemcpyDeviceToHost));
    orc_best_multiexp(curve, sc.data(), bases.data(), n, want);
    char msg[128];
    snprintf(msg, sizeof msg, "h2_msm_device n = %zu (curve %d) == oracle best_multiexp", n, curve);
    expect(same_point(curve, got, want), msg);
    CHECK_RC(p_h2_msm(curve, sc.data(), bases.data(), n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, got_h));
    snprintf(msg, sizeof msg, "h2_msm (host pointers) n = %zu (curve %d) == oracle best_multiexp", n, curve);
    expect(same_point(curve, got_h, want), msg);
    if (timing) {
        hipStream_t one = nullptr;                 // H2BENCH_MSM_STREAM=1: the one-at-a-time calls on a created stream (as a caller's would be) instead of the null stream
        if (getenv("H2BENCH_MSM_STREAM")) HIPCK(hipStreamCreateWithFlags(&one, hipStreamNonBlocking));
        for (int i = 0; i < 6; ++i) CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_o, one));
        HIPCK(hipDeviceSynchronize());
        const int R = 12;
        double t0 = now_ms();
        for (int i = 0; i < R; ++i) CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_o, one));
        const double enq_ms = (now_ms() - t0) / R;
        HIPCK(hipDeviceSynchronize());
        const double dev_ms = (now_ms() - t0) / R;
        printf("    host time to enqueue one call: %.4f ms\n", enq_ms);
        if (getenv("H2BENCH_CLOCK")) {      // shader clock / socket power with calls back to back for ~0.4 s, then as independent calls on 3 streams
            ClockWatch cw;
            cw.start();
            t0 = now_ms();
            int done = 0;
            while (now_ms() - t0 < 400.0) {
                for (int i = 0; i < 8; ++i) CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_o, nullptr));
                HIPCK(hipDeviceSynchronize());
                done += 8;
            }
            const double per = (now_ms() - t0) / done;
            printf("    one stream, back to back for 0.4 s: %.4f ms per call; %s\n", per, cw.finish().c_str());
        }
        {
            const int NS = 3;
            hipStream_t ss[NS];
            void *d_os[NS];
            for (int j = 0; j < NS; ++j) { HIPCK(hipStreamCreateWithFlags(&ss[j], hipStreamNonBlocking)); HIPCK(hipMalloc(&d_os[j], 96)); }
            for (int i = 0; i < 2 * NS; ++i) CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_os[i % NS], ss[i % NS]));
            HIPCK(hipDeviceSynchronize());
            const int RS = 6 * NS;
            ClockWatch cw;
            const bool watch = getenv("H2BENCH_CLOCK") != nullptr;
            double best = 1e30;
            for (int rep = 0; rep < (watch ? 12 : 2); ++rep) {
                if (watch && rep == 2) cw.start();
                t0 = now_ms();
                for (int i = 0; i < RS; ++i) CHECK_RC(p_h2_msm_device(curve, d_s, d_b, n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_os[i % NS], ss[i % NS]));
                HIPCK(hipDeviceSynchronize());
                best = std::min(best, (now_ms() - t0) / RS);
            }
            bool all_ok = true;
            for (int j = 0; j < NS; ++j) {
                uint64_t g3[12];
                HIPCK(hipMemcpy(g3, d_os[j], 96, hipMemcpyDeviceToHost));
                all_ok = all_ok && same_point(curve, g3, want);
            }
            snprintf(msg, sizeof msg, "h2_msm_device n = %zu as independent calls on %d streams == oracle best_multiexp", n, NS);
            expect(all_ok, msg);
            printf("    independent calls round-robin on %d streams: %.4f ms per call (%.1f M scalar-mults/s)%s%s\n", NS, best, n / best / 1e3, watch ? "; " : "",
                   watch ? cw.finish().c_str() : "");
            for (int j = 0; j < NS; ++j) { (void)hipStreamDestroy(ss[j]); (void)hipFree(d_os[j]); }
        }
        if (getenv("H2BENCH_MSM_DEVICE_ONLY")) {
            printf("generic best_multiexp n = %zu: %.4f ms device-resident (%.1f M scalar-mults/s)\n", n, dev_ms, n / dev_ms / 1e3);
            (void)hipFree(d_b); (void)hipFree(d_s); (void)hipFree(d_o);
            return;
        }
        for (int i = 0; i < 3; ++i) CHECK_RC(p_h2_msm(curve, sc.data(), bases.data(), n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, got_h));    // (workspaces, captured launch sequences)
        std::vector<double> hm;
        for (int i = 0; i < 9; ++i) {
            t0 = now_ms();
            CHECK_RC(p_h2_msm(curve, sc.data(), bases.data(), n, H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, got_h));
            hm.push_back(now_ms() - t0);
        }
        std::sort(hm.begin(), hm.end());
        const double host_ms = hm[hm.size() / 2];
        snprintf(msg, sizeof msg, "h2_msm (host pointers) n = %zu after the timed calls == oracle best_multiexp", n);
        expect(same_point(curve, got_h, want), msg);
        // the floor of that call: the same 96 bytes per point across PCIe and nothing else (pageable memory, as a caller's Vec is)
        t0 = now_ms();
        for (int i = 0; i < 5; ++i) {
            HIPCK(hipMemcpy(d_s, sc.data(), n * 32, hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(d_b, bases.data(), n * 64, hipMemcpyHostToDevice));
        }
        const double pcie_ms = (now_ms() - t0) / 5;
        printf("generic best_multiexp n = %zu: %.4f ms device-resident (%.1f M scalar-mults/s), %.4f ms from host pointers (median of 9 calls, PCIe inside; the two copies alone: %.4f ms)\n", n, dev_ms,
               n / dev_ms / 1e3, host_ms, pcie_ms);
    }
    (void)hipFree(d_b);
    (void)hipFree(d_s);
    (void)hipFree(d_o);
}

static void mode_msm(unsigned log_n, int curve) { mode_msm_n(((size_t)1 << log_n) + (getenv("H2BENCH_MSM_PLUS1") ? 1 : 0), curve, true); }

// ---- host-pointer seam --------------------------------------------------------------------------------------------------------
static void mode_host(unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    mode_msm(log_n, H2_PALLAS);
    std::vector<uint64_t> a(n * 4), omega(4);
    orc_random_field(H2_FP, 5, a.data(), n);
    orc_random_field(H2_FP, 6, omega.data(), 1);
    std::vector<uint64_t> want = a, got = a;
    orc_best_fft(H2_FP, want.data(), omega.data(), log_n);
    CHECK_RC(p_h2_ntt(H2_FP, got.data(), log_n, omega.data(), H2_FORM_MONTGOMERY));
    expect(got == want, "h2_ntt (host pointers) == oracle best_fft");
    double t0 = now_ms();
    for (int i = 0; i < 5; ++i) CHECK_RC(p_h2_ntt(H2_FP, got.data(), log_n, omega.data(), H2_FORM_MONTGOMERY));
    printf("h2_ntt 2^%u from host pointers: %.4f ms (H2D + passes + D2H)\n", log_n, (now_ms() - t0) / 5);
}

// ---- column-batched commits ---------------------------------------------------------------------------------------------------
static void mode_batch(unsigned log_n, int K, int curve) {
    const size_t n = (size_t)1 << log_n;
    const int sf = scalar_field(curve);
    uint64_t gen[8];
    generator(curve, gen);
    std::vector<uint64_t> bases(n * 8), w(8), blinds((size_t)K * 4);
    std::vector<std::vector<uint64_t>> cols(K, std::vector<uint64_t>(n * 4));
    orc_generate_bases(curve, gen, 0xBA7C4 + log_n, bases.data(), n);
    orc_generate_bases(curve, gen, 0x78, w.data(), 1);
    for (int c = 0; c < K; ++c) orc_random_field(sf, 2000 + c, cols[c].data(), n);
    if (K > 2) std::fill(cols[2].begin(), cols[2].end(), 0);            // an all-zero column inside the batch
    orc_random_field(sf, 0xB11E, blinds.data(), K);
    h2_bases_t g = 0;
    CHECK_RC(p_h2_bases_register_ex(curve, bases.data(), n, H2_FORM_MONTGOMERY, p_h2_commit_column_window_bits(n), &g));
    CHECK_RC(p_h2_bases_set_blind_base(g, w.data(), H2_FORM_MONTGOMERY));
    std::vector<void *> d_cols(K), d_bl(K), d_outs(K);
    void *d_blinds = nullptr, *d_out = nullptr;
    HIPCK(hipMalloc(&d_blinds, (size_t)K * 32));
    HIPCK(hipMalloc(&d_out, (size_t)K * 96));
    HIPCK(hipMemcpy(d_blinds, blinds.data(), (size_t)K * 32, hipMemcpyHostToDevice));
    for (int c = 0; c < K; ++c) {
        HIPCK(hipMalloc(&d_cols[c], n * 32));
        HIPCK(hipMemcpy(d_cols[c], cols[c].data(), n * 32, hipMemcpyHostToDevice));
        d_bl[c] = (char *)d_blinds + 32 * c;
        d_outs[c] = (char *)d_out + 96 * c;
    }
    auto batched = [&]() {
        return p_h2_commit_batch_device(g, d_cols.data(), (size_t)K, n, nullptr, d_bl.data(), H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_outs.data(), nullptr);
    };
    CHECK_RC(batched());
    HIPCK(hipDeviceSynchronize());
    std::vector<uint64_t> got((size_t)K * 12);
    HIPCK(hipMemcpy(got.data(), d_out, (size_t)K * 96, hipMemcpyDeviceToHost));
    bool all = true;
    for (int c = 0; c < K; ++c) {
        uint64_t want[12];
        orc_commit(curve, bases.data(), w.data(), cols[c].data(), blinds.data() + 4 * c, n, want);
        all = all && same_point(curve, got.data() + 12 * c, want);
    }
    char msg[128];
    snprintf(msg, sizeof msg, "h2_commit_batch_device: %d columns of 2^%u (curve %d, one all-zero column) == oracle Params::commit each", K, log_n, curve);
    expect(all, msg);
    for (int i = 0; i < 5; ++i) CHECK_RC(batched());
    HIPCK(hipDeviceSynchronize());
    const int R = 20;
    double t0 = now_ms();
    for (int i = 0; i < R; ++i) CHECK_RC(batched());
    HIPCK(hipDeviceSynchronize());
    const double b_ms = (now_ms() - t0) / R;
    t0 = now_ms();
    for (int i = 0; i < R; ++i)
        for (int c = 0; c < K; ++c) CHECK_RC(p_h2_commit_device(g, d_cols[c], n, nullptr, d_bl[c], H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, d_outs[c], nullptr));
    HIPCK(hipDeviceSynchronize());
    const double s_ms = (now_ms() - t0) / R;
    printf("batch: %d columns of 2^%u in one call %.4f ms; the same as %d single commits on one stream %.4f ms\n", K, log_n, b_ms, K, s_ms);
    for (void *p : d_cols) (void)hipFree(p);
    (void)hipFree(d_blinds);
    (void)hipFree(d_out);
    p_h2_bases_free(g);
}

// ---- EvaluationDomain transforms (poly/domain.rs:227-255, 303-325, 375-383) ---------------------------------------------------------
// The constants (omega, zeta, divisors) are RANDOM field elements here: the transforms are the same formulas on both sides for any
// values (the parity contract for best_fft already holds for any omega), and the driver needs no field arithmetic of its own.
static void mode_domain(unsigned k, unsigned ext, int field) {
    const unsigned ek = k + ext;
    const size_t n = (size_t)1 << k, ne = (size_t)1 << ek;
    std::vector<uint64_t> a(n * 4), c(5 * 4);
    orc_random_field(field, 41 + k, a.data(), n);
    orc_random_field(field, 42 + k, c.data(), 5);
    const uint64_t *omega = c.data(), *divisor = c.data() + 4, *zeta = c.data() + 8, *zeta_inv = c.data() + 12, *ext_omega = c.data() + 16;
    void *d_a = nullptr, *d_e = nullptr;
    HIPCK(hipMalloc(&d_a, n * 32));
    HIPCK(hipMalloc(&d_e, ne * 32));
    // ifft
    HIPCK(hipMemcpy(d_a, a.data(), n * 32, hipMemcpyHostToDevice));
    CHECK_RC(p_h2_ifft_device(field, d_a, k, omega, divisor, H2_FORM_MONTGOMERY, nullptr));
    HIPCK(hipDeviceSynchronize());
    std::vector<uint64_t> coeff(n * 4), want = a;
    HIPCK(hipMemcpy(coeff.data(), d_a, n * 32, hipMemcpyDeviceToHost));
    orc_ifft(field, want.data(), omega, k, divisor);
    expect(coeff == want, "h2_ifft_device == oracle ifft at every index");
    // coeff_to_extended
    CHECK_RC(p_h2_coeff_to_extended_device(field, d_a, d_e, k, ek, zeta, zeta_inv, ext_omega, H2_FORM_MONTGOMERY, nullptr));
    HIPCK(hipDeviceSynchronize());
    std::vector<uint64_t> extv(ne * 4), ext_want(ne * 4, 0);
    HIPCK(hipMemcpy(extv.data(), d_e, ne * 32, hipMemcpyDeviceToHost));
    memcpy(ext_want.data(), want.data(), n * 32);
    orc_coeff_to_extended(field, ext_want.data(), k, ek, zeta, zeta_inv, ext_omega);
    expect(extv == ext_want, "h2_coeff_to_extended_device == oracle coeff_to_extended at every index");
    // extended_to_coeff of a generic extended vector
    std::vector<uint64_t> e(ne * 4);
    orc_random_field(field, 43 + k, e.data(), ne);
    HIPCK(hipMemcpy(d_e, e.data(), ne * 32, hipMemcpyHostToDevice));
    CHECK_RC(p_h2_extended_to_coeff_device(field, d_e, ek, zeta, zeta_inv, omega, divisor, H2_FORM_MONTGOMERY, nullptr));
    HIPCK(hipDeviceSynchronize());
    std::vector<uint64_t> back(ne * 4);
    HIPCK(hipMemcpy(back.data(), d_e, ne * 32, hipMemcpyDeviceToHost));
    orc_extended_to_coeff(field, e.data(), ek, zeta, zeta_inv, omega, divisor);
    expect(back == e, "h2_extended_to_coeff_device == oracle extended_to_coeff at every index");
    // warm timings
    auto time_it = [&](const char *what, auto fn) {
        for (int i = 0; i < 10; ++i) fn();
        (void)hipDeviceSynchronize();
        const int R = 30;
        const double t0 = now_ms();
        for (int i = 0; i < R; ++i) fn();
        (void)hipDeviceSynchronize();
        printf("%s: %.4f ms\n", what, (now_ms() - t0) / R);
    };
    char lbl[96];
    snprintf(lbl, sizeof lbl, "ifft 2^%u", k);
    time_it(lbl, [&] { p_h2_ifft_device(field, d_a, k, omega, divisor, H2_FORM_MONTGOMERY, nullptr); });
    snprintf(lbl, sizeof lbl, "coeff_to_extended 2^%u -> 2^%u", k, ek);
    time_it(lbl, [&] { p_h2_coeff_to_extended_device(field, d_a, d_e, k, ek, zeta, zeta_inv, ext_omega, H2_FORM_MONTGOMERY, nullptr); });
    snprintf(lbl, sizeof lbl, "extended_to_coeff 2^%u", ek);
    time_it(lbl, [&] { p_h2_extended_to_coeff_device(field, d_e, ek, zeta, zeta_inv, omega, divisor, H2_FORM_MONTGOMERY, nullptr); });
    (void)hipFree(d_a);
    (void)hipFree(d_e);
}

static std::vector<unsigned> parse_list(const char *s) {
    std::vector<unsigned> v;
    for (const char *p = s; *p;) {
        v.push_back((unsigned)strtoul(p, (char **)&p, 10));
        if (*p == ',') ++p;
    }
    return v;
}

int main(int argc, char **argv) {
    if (!load_library(argv[0])) return 2;
    if (p_h2_device_count() <= 0) { printf("no GPU: h2bench needs an MI355X\n"); return 2; }
    if (p_h2_init(0) != H2_OK) { printf("h2_init: %s\n", p_h2_last_error()); return 2; }
    if (const char *e = getenv("H2BENCH_EXTRA_STREAMS")) {      // a process with many streams open (as a prover has): HIP multiplexes them onto few hardware queues
        for (int i = 0; i < atoi(e); ++i) {
            hipStream_t s;
            void *d = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d, 256) == hipSuccess) (void)hipMemsetAsync(d, 0, 256, s);
        }
        (void)hipDeviceSynchronize();
    }
    const std::string mode = argc > 1 ? argv[1] : "commit";
    auto arg = [&](int i, long dflt) { return argc > i ? atol(argv[i]) : dflt; };
    if (mode == "commit") mode_commit((unsigned)arg(2, 20), parse_list(argc > 3 ? argv[3] : "20"), (int)arg(4, 5), parse_list(argc > 5 ? argv[5] : "3"), (int)arg(6, H2_PALLAS), (int)arg(7, 1));
    else if (mode == "ntt") mode_ntt(parse_list(argc > 2 ? argv[2] : "16,18,20,22"), (int)arg(3, H2_FP), arg(4, 1) != 0);
    else if (mode == "msm") mode_msm((unsigned)arg(2, 20), (int)arg(3, H2_PALLAS));
    else if (mode == "host") mode_host((unsigned)arg(2, 20));
    else if (mode == "batch") mode_batch((unsigned)arg(2, 13), (int)arg(3, 8), (int)arg(4, H2_PALLAS));
    else if (mode == "domain") mode_domain((unsigned)arg(2, 20), (unsigned)arg(3, 1), (int)arg(4, H2_FP));
    else if (mode == "parity") {
        for (int curve : {H2_PALLAS, H2_VESTA})
            for (size_t n : {(size_t)0, (size_t)1, (size_t)2, (size_t)255, (size_t)4097, (size_t)65535, (size_t)65536, (size_t)65537, (size_t)300001})
                mode_msm_n(n, curve, false);         // (n = 0: the identity)
        for (int field : {H2_FP, H2_FQ}) mode_ntt({1, 2, 5, 10, 11, 13, 16, 19, 20, 21}, field, true);
        mode_commit(14, {8}, 2, {3}, H2_VESTA);
        mode_commit(18, {8}, 2, {3}, H2_PALLAS);
        mode_batch(12, 8, H2_VESTA);
        mode_batch(16, 3, H2_PALLAS);
        mode_domain(12, 2, H2_FQ);
        mode_domain(17, 1, H2_FP);
    } else {
        printf("usage: h2bench commit|ntt|msm|host|batch|domain|parity ... (see the head of bench/native/h2bench.cpp)\n");
        return 2;
    }
    printf(g_fail ? "H2BENCH FAIL (%d)\n" : "H2BENCH OK\n", g_fail);
    return g_fail ? 1 : 0;
}
