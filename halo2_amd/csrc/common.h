// Host-side plumbing shared by the MSM / NTT translation units: status codes, HIP error mapping,
// a tiny size-keyed device workspace cache.  No torch, no third-party types.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <memory>
#include <mutex>
#include <utility>

#include "../../include/halo2_mi355x.h"

namespace h2 {

#define H2_HIP(expr)                                      \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) {                           \
            h2::set_last_hip_error(_e, __FILE__, __LINE__); \
            return H2_ERR_HIP;                            \
        }                                                 \
    } while (0)

void set_last_hip_error(hipError_t e, const char *file, int line);
void set_last_error_msg(const char *msg);   // what h2_last_error() returns on this thread

// The laboratory is not in the product.  Every A/B arm and sweep knob of the experiments behind DESIGN_LOG.md (window widths, sort
// geometries, plan depths, pipeline shapes ...) is read through ab_env(): in the shipped library it is a constant null -- each arm folds
// to its default at compile time and NO environment variable changes what the prover computes or how.  `make ab` builds the same sources
// with -DH2_AB=1 into build/ab/libhalo2_mi355x_ab.so, where the switches are live; the A/B parity tests and bench/tools point at that
// build (H2_LIB_PATH / H2BENCH_LIB).  The only variable the shipped library reads is the diagnostic H2_TIMELINE (include/halo2_mi355x.h).
#ifdef H2_AB
inline const char *ab_env(const char *name) { return getenv(name); }
#else
inline const char *ab_env(const char *) { return nullptr; }
#endif

// Grow-only device buffer.  One instance per use site, guarded by the owning context's mutex.
// counts every (re)allocation and release of a DevBuf in the process: a captured hipGraph names device pointers, and is replayed
// only while the count it was captured under still stands (msm.hip, the range pipeline of h2_msm)
unsigned long devbuf_epoch();
void devbuf_epoch_bump();
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return H2_OK;
        devbuf_epoch_bump();
        if (ptr) {
            hipError_t e = hipFree(ptr);
            ptr = nullptr;
            cap = 0;
            if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) { ptr = nullptr; set_last_hip_error(e, __FILE__, __LINE__); return H2_ERR_HIP; }
        cap = want;
        return H2_OK;
    }
    void release() {
        if (ptr) {
            devbuf_epoch_bump();
            (void)hipFree(ptr);
        }
        ptr = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

// One workspace per (device, stream): calls enqueued on different streams never share scratch (a mutex inside T only
// serialises host-side enqueueing).  T needs `std::mutex mu` and `void release_all()`.
template <typename T> struct StreamContexts {
    std::mutex mu;
    std::map<std::pair<int, hipStream_t>, std::unique_ptr<T>> ctxs;
    T &get(hipStream_t st) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        auto &slot = ctxs[std::make_pair(dev, st)];
        if (!slot) slot.reset(new T());
        return *slot;
    }
    void release_current_device() {   // h2_trim
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : ctxs) {
            if (kv.first.first != dev) continue;
            std::lock_guard<std::mutex> cl(kv.second->mu);
            kv.second->release_all();
        }
    }
};

// Optional kernel timing with HIP events on the launching stream (h2_profile_enable): bench.py uses it to
// measure the dominant kernels' average duration live, inside the timed region.
enum ProfSlot { PROF_MSM_ACCUMULATE = 0, PROF_NTT_PASS = 1, PROF_MSM_SORT = 2, PROF_MSM_REDUCE = 3, PROF_SLOTS = 4 };
bool prof_enabled();
void prof_begin(int slot, hipStream_t st);
void prof_end(int slot, hipStream_t st);

// Device table omega^0 .. omega^(2^(L-1) - 1) (Montgomery, 32 B each) from the NTT's per-(field, omega, log n) cache
// (ntt.hip); built on `st` when missing.  Shared with the curve-point FFT (ecfft.hip).
int ntt_twiddle_table(int field, int L, const uint64_t omega_mont[4], hipStream_t st, const uint32_t **d_tw);

// h2_trim: each translation unit hands its cached device scratch of the current device back to the allocator
void msm_release_workspaces();
void ntt_release_workspaces();
void poly_release_workspaces();
void ipa_release_workspaces();
void eval_release_workspaces();
void lookup_release_workspaces();

// Confirms a usable gfx950 device exists; every entry point calls this first so a missing GPU or
// runtime fails loudly (H2_ERR_NODEV) instead of silently doing nothing.
int ensure_device();

}  // namespace h2
