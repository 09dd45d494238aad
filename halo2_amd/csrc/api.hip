// Library-level entry points of libhalo2_mi355x.so: device discovery, error reporting.
// The MSM entry points live in msm.hip, the NTT entry points in ntt.hip (see include/halo2_mi355x.h).
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include <algorithm>
#include <atomic>

#include "common.h"

namespace h2 {

static thread_local char g_err[256] = "no error";

void set_last_hip_error(hipError_t e, const char *file, int line) {
    const char *base = strrchr(file, '/');
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), base ? base + 1 : file, line);
}

void set_last_error_msg(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

int ensure_device() {
    static thread_local int checked = 0;
    if (checked) return H2_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        snprintf(g_err, sizeof g_err, "no HIP device available (%s): the MI355X path has no CPU fallback",
                 e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return H2_ERR_NODEV;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        snprintf(g_err, sizeof g_err, "cannot query the current HIP device");
        return H2_ERR_NODEV;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof g_err, "device %d is %s; this library ships gfx950 (MI355X) code only", dev, prop.gcnArchName);
        return H2_ERR_NODEV;
    }
    checked = 1;
    return H2_OK;
}

static std::atomic<unsigned long> g_devbuf_epoch{0};
unsigned long devbuf_epoch() { return g_devbuf_epoch.load(std::memory_order_acquire); }
void devbuf_epoch_bump() { g_devbuf_epoch.fetch_add(1, std::memory_order_acq_rel); }

// ---- event profiler ---------------------------------------------------------------------------
struct ProfState {
    std::mutex mu;
    bool on = false;
    unsigned mask = 0xF;        // slots that record (h2_profile_enable(2): the dominant kernels only -- every pair of events costs the stream ~10 us)
    struct Pair { hipEvent_t a, b; };
    std::vector<Pair> pending[PROF_SLOTS];
    hipEvent_t open[PROF_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    double total_ms[PROF_SLOTS] = {0, 0, 0, 0};
    double busy_ms[PROF_SLOTS] = {0, 0, 0, 0};     // union of the launch intervals drained so far
    uint64_t count[PROF_SLOTS] = {0, 0, 0, 0};
};
static ProfState g_prof;
bool prof_enabled() { return g_prof.on; }
void prof_begin(int slot, hipStream_t st) {
    if (!g_prof.on || !((g_prof.mask >> slot) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_prof.open[slot] = e;
}
void prof_end(int slot, hipStream_t st) {
    if (!g_prof.on || !((g_prof.mask >> slot) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.open[slot]) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_prof.pending[slot].push_back({g_prof.open[slot], e});
    g_prof.open[slot] = nullptr;
}

}  // namespace h2

extern "C" int h2_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(h2::g_prof.mu);
    h2::g_prof.on = on != 0;
    h2::g_prof.mask = on == 2 ? ((1u << h2::PROF_MSM_ACCUMULATE) | (1u << h2::PROF_NTT_PASS)) : 0xFu;
    if (on) {
        for (int s = 0; s < h2::PROF_SLOTS; ++s) {
            h2::g_prof.total_ms[s] = 0;
            h2::g_prof.busy_ms[s] = 0;
            h2::g_prof.count[s] = 0;
        }
    }
    return H2_OK;
}

// drains the pending event pairs of a slot: sum of the launch durations, and the length of the UNION of the launch intervals
// (intervals are placed on one clock by measuring every event against the first one drained: hipEventElapsedTime works
// across streams of a device).  When launches issued on several streams overlap, the union is the time the device spent on
// this kernel; the sum counts overlapped time once per launch.
static void prof_drain(int slot) {
    auto &pend = h2::g_prof.pending[slot];
    if (pend.empty()) return;
    std::vector<std::pair<double, double>> iv;
    hipEvent_t base = pend.front().a;
    for (auto &pr : pend) {
        float ms = 0, t0 = 0;
        if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
            h2::g_prof.total_ms[slot] += ms;
            h2::g_prof.count[slot] += 1;
            if (pr.a == base || hipEventElapsedTime(&t0, base, pr.a) == hipSuccess) iv.push_back({(double)t0, (double)t0 + ms});
        }
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0, lo = 0, hi = -1;
    for (auto &x : iv) {
        if (hi < lo || x.first > hi) {
            if (hi >= lo) busy += hi - lo;
            lo = x.first;
            hi = x.second;
        } else if (x.second > hi) hi = x.second;
    }
    if (hi >= lo) busy += hi - lo;
    h2::g_prof.busy_ms[slot] += busy;
    for (auto &pr : pend) {
        (void)hipEventDestroy(pr.a);
        (void)hipEventDestroy(pr.b);
    }
    pend.clear();
}

extern "C" int h2_profile_read(int slot, double *total_ms, uint64_t *launches) {
    if (slot < 0 || slot >= h2::PROF_SLOTS || !total_ms || !launches) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(h2::g_prof.mu);
    prof_drain(slot);
    *total_ms = h2::g_prof.total_ms[slot];
    *launches = h2::g_prof.count[slot];
    return H2_OK;
}

extern "C" int h2_profile_read_busy(int slot, double *total_ms, double *busy_ms, uint64_t *launches) {
    if (slot < 0 || slot >= h2::PROF_SLOTS || !total_ms || !busy_ms || !launches) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(h2::g_prof.mu);
    prof_drain(slot);
    *total_ms = h2::g_prof.total_ms[slot];
    *busy_ms = h2::g_prof.busy_ms[slot];
    *launches = h2::g_prof.count[slot];
    return H2_OK;
}

extern "C" int h2_current_device(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return dev;
}

extern "C" int h2_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

extern "C" int h2_init(int device) {
    int count = h2_device_count();
    if (count <= 0) {
        snprintf(h2::g_err, sizeof h2::g_err, "no HIP device available");
        return H2_ERR_NODEV;
    }
    if (device < 0 || device >= count) return H2_ERR_ARGS;
    H2_HIP(hipSetDevice(device));
    return h2::ensure_device();
}

extern "C" const char *h2_last_error(void) { return h2::g_err; }

extern "C" int h2_trim(void) {
    int rc = h2::ensure_device();
    if (rc != H2_OK) return rc;
    H2_HIP(hipDeviceSynchronize());
    h2::msm_release_workspaces();
    h2::ntt_release_workspaces();
    h2::poly_release_workspaces();
    h2::ipa_release_workspaces();
    h2::eval_release_workspaces();
    h2::lookup_release_workspaces();
    return H2_OK;
}
