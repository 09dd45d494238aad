// Multi-GPU entry points of the C ABI: the reference's `create_proof` is ONE process
// (plonk/prover.rs:35-724), so the drop-in seam for a node with several MI355X is a call that takes the whole phase and
// spreads it over the devices itself:
//
//   h2_commit_batch_multi   the independent column commits of a prover phase (plonk/prover.rs:93-101, 301-313;
//                           vanishing/prover.rs:96-108), column i on device i mod ndev, one host thread and a few streams
//                           per device, each device holding its own registered copy of the bases.  No collective: a
//                           commit's result is one 96-byte point that goes back to the host.
//   h2_msm_split_multi      ONE multiexp cut into ndev contiguous point ranges (a single large commit, an opening-argument
//                           round), partial sums brought to the host and added: RCCL cannot reduce curve points.
//   h2_rccl_* / h2_msm_split_rccl_device
//                           the same split for the one-process-per-GPU model (torch.distributed / MPI launchers): every rank
//                           computes its range's partial on its own GPU, ONE ncclAllGather of 96 bytes per rank over
//                           xGMI, then every rank adds the `world` partials locally.  RCCL is bound at run time
//                           (dlopen "librccl.so"), so the library has no load-time dependency on it and does not clash with
//                           the copy a host framework may already have loaded.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

typedef uint32_t u32;

namespace h2 {
namespace {

constexpr int kStreamsPerDevice = 3;

// per-device staging for h2_commit_batch_multi: scalar columns, blinds, w, outputs
struct DeviceLane {
    std::mutex mu;
    int device = -1;
    hipStream_t st[kStreamsPerDevice] = {nullptr, nullptr, nullptr};
    void *d_scalars[kStreamsPerDevice] = {nullptr, nullptr, nullptr};
    size_t cap[kStreamsPerDevice] = {0, 0, 0};
    void *d_small = nullptr;     // [w 64 B][per stream: blind 32 B, out 96 B]
    int prepare(int dev, size_t bytes) {
        if (device != dev) {
            device = dev;
            for (int i = 0; i < kStreamsPerDevice; ++i) H2_HIP(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            H2_HIP(hipMalloc(&d_small, 64 + kStreamsPerDevice * 128));
        }
        for (int i = 0; i < kStreamsPerDevice; ++i)
            if (cap[i] < bytes) {
                if (d_scalars[i]) H2_HIP(hipFree(d_scalars[i]));
                d_scalars[i] = nullptr;
                cap[i] = 0;
                H2_HIP(hipMalloc(&d_scalars[i], bytes));
                cap[i] = bytes;
            }
        return H2_OK;
    }
};
DeviceLane &lane_of(int dev) {
    static DeviceLane lanes[64];
    return lanes[dev & 63];
}

int run_on_devices(int ndev, const std::function<int(int)> &body) {
    std::vector<int> rcs((size_t)ndev, H2_OK);
    if (ndev == 1) return body(0);
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; ++d) th.emplace_back([&, d] { rcs[(size_t)d] = body(d); });
    for (auto &t : th) t.join();
    for (int rc : rcs)
        if (rc != H2_OK) return rc;
    return H2_OK;
}

}  // namespace
}  // namespace h2

using namespace h2;

extern "C" int h2_commit_batch_multi(const h2_bases_t *handles, const int *devices, int ndev, const uint64_t *const *scalars, size_t count,
                                     size_t n, const uint64_t *w_xy, const uint64_t *const *blinds, int form, int out_kind,
                                     uint64_t *const *outs) {
    if (!handles || !devices || ndev <= 0 || ndev > 64 || !scalars || !outs || (w_xy && !blinds) ||
        (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE))
        return H2_ERR_ARGS;
    if (count == 0) return H2_OK;
    int have = h2_device_count();
    for (int d = 0; d < ndev; ++d)
        if (devices[d] < 0 || devices[d] >= have) return H2_ERR_ARGS;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const size_t out_bytes = out_kind == H2_OUT_AFFINE ? 64 : 96;
    int rc = run_on_devices(ndev, [&](int d) -> int {
        H2_HIP(hipSetDevice(devices[d]));
        int r = ensure_device();
        if (r != H2_OK) return r;
        DeviceLane &L = lane_of(devices[d]);
        std::lock_guard<std::mutex> lk(L.mu);
        if ((r = L.prepare(devices[d], std::max<size_t>(n, 1) * 32)) != H2_OK) return r;
        char *small = (char *)L.d_small;
        // the blind base belongs to the handle and is compared by CONTENT: a call with another w installs it on this device first
        if (w_xy && (r = h2_bases_set_blind_base(handles[d], w_xy, form)) != H2_OK) return r;
        size_t k = 0;
        for (size_t i = (size_t)d; i < count; i += (size_t)ndev, ++k) {
            const int s = (int)(k % kStreamsPerDevice);
            hipStream_t st = L.st[s];
            char *bl = small + 64 + s * 128, *out = bl + 32;
            // the stream's staging slot is free again once its previous commit's result has been copied out (same stream: ordered)
            if (n) H2_HIP(hipMemcpyAsync(L.d_scalars[s], scalars[i], n * 32, hipMemcpyHostToDevice, st));
            if (w_xy) H2_HIP(hipMemcpyAsync(bl, blinds[i], 32, hipMemcpyHostToDevice, st));
            r = h2_commit_device(handles[d], L.d_scalars[s], n, nullptr, w_xy ? bl : nullptr, form, out_kind, out, st);
            if (r != H2_OK) return r;
            H2_HIP(hipMemcpyAsync(outs[i], out, out_bytes, hipMemcpyDeviceToHost, st));
        }
        for (int s = 0; s < kStreamsPerDevice; ++s) H2_HIP(hipStreamSynchronize(L.st[s]));
        return H2_OK;
    });
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int h2_msm_split_multi(int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, const int *devices, int ndev,
                                  int form, int out_kind, uint64_t *out) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || !devices || ndev <= 0 || ndev > 64 || !out || (n && (!scalars || !bases_xy)))
        return H2_ERR_ARGS;
    int have = h2_device_count();
    for (int d = 0; d < ndev; ++d)
        if (devices[d] < 0 || devices[d] >= have) return H2_ERR_ARGS;
    int prev = 0;
    (void)hipGetDevice(&prev);
    std::vector<uint64_t> partial((size_t)ndev * 12, 0);
    int rc = run_on_devices(ndev, [&](int d) -> int {
        H2_HIP(hipSetDevice(devices[d]));
        const size_t lo = n * (size_t)d / (size_t)ndev, hi = n * (size_t)(d + 1) / (size_t)ndev;      // contiguous ranges
        return h2_msm(curve, scalars + 4 * lo, bases_xy + 8 * lo, hi - lo, form, H2_OUT_JACOBIAN, &partial[(size_t)d * 12]);
    });
    if (rc == H2_OK) {
        // the one exchange step: ndev x 96 bytes to one device, added there
        (void)hipSetDevice(devices[0]);
        void *d_tmp = nullptr;
        hipError_t e = hipMalloc(&d_tmp, (size_t)(ndev + 1) * 96);
        if (e == hipSuccess) e = hipMemcpy(d_tmp, partial.data(), (size_t)ndev * 96, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            char *d_res = (char *)d_tmp + (size_t)ndev * 96;
            rc = h2_points_sum_device(curve, d_tmp, (size_t)ndev, form, out_kind, d_res, nullptr);
            if (rc == H2_OK) e = hipMemcpy(out, d_res, out_kind == H2_OUT_AFFINE ? 64 : 96, hipMemcpyDeviceToHost);
        }
        if (d_tmp) (void)hipFree(d_tmp);
        if (e != hipSuccess) { set_last_hip_error(e, __FILE__, __LINE__); rc = H2_ERR_HIP; }
    }
    (void)hipSetDevice(prev);
    return rc;
}

// ---- RCCL, bound at run time -------------------------------------------------------------------------------------------
namespace {
struct ncclUniqueIdBytes { char internal[128]; };
typedef void *ncclComm_t;
struct Rccl {
    std::mutex mu;
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueIdBytes *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueIdBytes, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    // exchange buffer: [world + 1] slots of kSlot bytes -- a Jacobian point (96 B) and a STATUS word behind it (0 = this rank's partial is
    // good); slot `world` = this rank's payload.  Behind the slots: [world] packed points (96 B each) for the sum, then one word that
    // counts the failed ranks.  A rank whose local work failed still enters the all-gather (its peers would wait for ever) with its
    // status set, and EVERY rank then returns an error: a peer's HIP error must not become a plausible-looking wrong commitment.
    static constexpr size_t kSlot = 128;
    void *d_buf = nullptr;
    u32 *h_failed = nullptr;   // pinned: the failed-rank count read back after the exchange
    size_t packed_off() const { return (size_t)(world + 1) * kSlot; }
    size_t failed_off() const { return packed_off() + (size_t)world * 96; }
    int load() {
        if (lib) return H2_OK;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names)
            if ((lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) {
            char msg[400];
            snprintf(msg, sizeof msg, "RCCL is not available: %s", dlerror());
            set_last_error_msg(msg);
            return H2_ERR_NODEV;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) {
            set_last_error_msg("librccl.so lacks an expected symbol");
            return H2_ERR_NODEV;
        }
        return H2_OK;
    }
    int fail(int code, const char *what) {
        char msg[400];
        snprintf(msg, sizeof msg, "%s: RCCL error %d (%s)", what, code, GetErrorString ? GetErrorString(code) : "?");
        set_last_error_msg(msg);
        return H2_ERR_HIP;
    }
};
Rccl g_rccl;
}  // namespace

extern "C" int h2_rccl_unique_id(uint8_t id_out[128]) {
    if (!id_out) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    int rc = g_rccl.load();
    if (rc != H2_OK) return rc;
    ncclUniqueIdBytes id;
    int e = g_rccl.GetUniqueId(&id);
    if (e) return g_rccl.fail(e, "ncclGetUniqueId");
    memcpy(id_out, id.internal, 128);
    return H2_OK;
}

extern "C" int h2_rccl_init(const uint8_t id[128], int rank, int world) {
    if (!id || world <= 0 || rank < 0 || rank >= world) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    if ((rc = g_rccl.load()) != H2_OK) return rc;
    if (g_rccl.comm) return H2_ERR_ARGS;     // one communicator per process (one process per GPU)
    ncclUniqueIdBytes u;
    memcpy(u.internal, id, 128);
    int e = g_rccl.CommInitRank(&g_rccl.comm, world, u, rank);
    if (e) {
        g_rccl.comm = nullptr;
        return g_rccl.fail(e, "ncclCommInitRank");
    }
    g_rccl.rank = rank;
    g_rccl.world = world;
    H2_HIP(hipMalloc(&g_rccl.d_buf, g_rccl.failed_off() + 16));
    H2_HIP(hipHostMalloc((void **)&g_rccl.h_failed, 16, hipHostMallocDefault));
    return H2_OK;
}

extern "C" int h2_rccl_finalize(void) {
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    if (g_rccl.comm) {
        (void)hipDeviceSynchronize();
        (void)g_rccl.CommDestroy(g_rccl.comm);
        g_rccl.comm = nullptr;
        if (g_rccl.d_buf) (void)hipFree(g_rccl.d_buf);
        g_rccl.d_buf = nullptr;
        if (g_rccl.h_failed) (void)hipHostFree(g_rccl.h_failed);
        g_rccl.h_failed = nullptr;
    }
    return H2_OK;
}

// after the all-gather: slot r (kSlot bytes: point + status) -> packed point r, and the number of ranks whose status is set
__global__ void rccl_unpack_slots(const u32 *__restrict__ slots, u32 *__restrict__ packed, u32 *__restrict__ failed, int world, u32 slot_words) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u32 bad = 0;
        for (int r = 0; r < world; ++r) bad += slots[(size_t)r * slot_words + 24] != 0u;
        *failed = bad;
    }
    for (u32 i = threadIdx.x; i < (u32)world * 24u; i += blockDim.x) packed[i] = slots[(size_t)(i / 24u) * slot_words + i % 24u];
}
namespace {
// `mine` (slot `world`) holds this rank's partial, written by work enqueued on `st` that returned local_rc.  Sets the status word,
// all-gathers the slots, unpacks, reads the failed-rank count back (ONE stream synchronisation: the price of never returning a
// wrong point) and adds the partials.  Every rank returns an error when any rank failed.
int rccl_exchange_and_sum(int curve, int local_rc, int form, int out_kind, void *d_out, hipStream_t st, const char *what) {
    Rccl &R = g_rccl;
    char *buf = (char *)R.d_buf, *mine = buf + (size_t)R.world * Rccl::kSlot;
    const u32 status = local_rc == H2_OK ? 0u : 1u;
    if (status) (void)hipMemsetAsync(mine, 0, 96, st);
    // the status word is SET ON THE DEVICE (a memset node carries its value): no host source that would have to outlive this call --
    // on the failure path the function returns right after the collective is enqueued
    hipError_t he = hipMemsetAsync(mine + 96, status ? 0xFF : 0, 4, st);
    int e = R.AllGather(mine, buf, Rccl::kSlot, /*ncclChar*/ 0, R.comm, st);
    if (local_rc != H2_OK) return local_rc;                 // the peers learn of it from the status word
    if (he != hipSuccess) { set_last_hip_error(he, __FILE__, __LINE__); return H2_ERR_HIP; }
    if (e) return R.fail(e, "ncclAllGather");
    u32 *packed = (u32 *)(buf + R.packed_off()), *failed = (u32 *)(buf + R.failed_off());
    hipLaunchKernelGGL(rccl_unpack_slots, dim3(1), dim3(256), 0, st, (const u32 *)buf, packed, failed, R.world, (u32)(Rccl::kSlot / 4));
    H2_HIP(hipGetLastError());
    H2_HIP(hipMemcpyAsync(R.h_failed, failed, 4, hipMemcpyDeviceToHost, st));
    H2_HIP(hipStreamSynchronize(st));
    if (*R.h_failed) {
        char msg[200];
        snprintf(msg, sizeof msg, "%s: %u of %d ranks failed their range; no result (a partial sum would be a wrong point)", what, *R.h_failed, R.world);
        set_last_error_msg(msg);
        return H2_ERR_PEER;
    }
    return h2_points_sum_device(curve, packed, (size_t)R.world, form, out_kind, d_out, st);
}
}  // namespace

// Every rank passes the WHOLE problem's device arrays (or at least its own range at the right offsets): rank r multiplies
// points [n r / world, n (r + 1) / world), the partials are all-gathered (a 128-byte slot per rank: 96-byte point + status), and
// every rank writes the total.
extern "C" int h2_msm_split_rccl_device(int curve, const void *d_scalars, const void *d_bases_xy, size_t n, int form, int out_kind,
                                        void *d_out, void *stream) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || !d_out || (n && (!d_scalars || !d_bases_xy))) return H2_ERR_ARGS;
    if ((form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE)) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    if (!g_rccl.comm) return H2_ERR_HANDLE;
    hipStream_t st = (hipStream_t)stream;
    const int world = g_rccl.world, rank = g_rccl.rank;
    const size_t lo = n * (size_t)rank / (size_t)world, hi = n * (size_t)(rank + 1) / (size_t)world;
    char *mine = (char *)g_rccl.d_buf + (size_t)world * Rccl::kSlot;
    // partial in Montgomery Jacobian form whatever the caller's form is: the sum kernel below reads Montgomery limbs
    const int rc = h2_msm_device(curve, (const char *)d_scalars + 32 * lo, (const char *)d_bases_xy + 64 * lo, hi - lo, form, H2_OUT_JACOBIAN, mine, st);
    // (argument errors above are rank-independent: every rank returns before the exchange; from here on a failure is rank-LOCAL and
    // the rank still takes part)
    return rccl_exchange_and_sum(curve, rc, form, out_kind, d_out, st, "h2_msm_split_rccl_device");
}

// The same exchange for a commit over REGISTERED bases (Params::commit, an opening-argument round): every rank holds the table
// (h2_bases_register on its GPU) and the column; rank r commits table columns [n r / world, n (r + 1) / world) -- no doubling
// chain, no per-call base traffic -- the last rank carries the blind term, then ONE all-gather and the local sum.
extern "C" int h2_commit_split_rccl_device(h2_bases_t g, const void *d_scalars, size_t n, const void *d_blind, int form, int out_kind,
                                           void *d_out, void *stream) {
    if (!d_out || (n && !d_scalars)) return H2_ERR_ARGS;
    size_t have = 0;
    int curve = 0;
    int rc = h2_bases_info(g, &have, nullptr, &curve);
    if (rc != H2_OK) return rc;
    if (n > have) return H2_ERR_ARGS;
    // rank-independent preconditions are checked on EVERY rank before any work: only the last rank hands the blind to its range
    // commit, and a failure there alone would leave the other ranks waiting in the all-gather
    if ((form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || (out_kind != H2_OUT_JACOBIAN && out_kind != H2_OUT_AFFINE)) return H2_ERR_ARGS;
    if (d_blind && h2_bases_blind_base_set(g) != 1) {
        set_last_error_msg("split commit with a blind but the handle has no blind base: call h2_bases_set_blind_base on every rank");
        return H2_ERR_ARGS;
    }
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    if (!g_rccl.comm) return H2_ERR_HANDLE;
    hipStream_t st = (hipStream_t)stream;
    const int world = g_rccl.world, rank = g_rccl.rank;
    const size_t lo = n * (size_t)rank / (size_t)world, hi = n * (size_t)(rank + 1) / (size_t)world;
    char *mine = (char *)g_rccl.d_buf + (size_t)world * Rccl::kSlot;
    rc = h2_commit_range_device(g, (const char *)d_scalars + 32 * lo, lo, hi - lo, rank == world - 1 ? d_blind : nullptr, form, H2_OUT_JACOBIAN,
                                mine, st);
    // fault injection for the tests (H2_TEST_FAIL_RANK=r: rank r reports a local failure after its range commit): every rank must
    // come back with an error and nobody may hang.  Read through ab_env(): compiled OUT of the shipped library (common.h), live in the
    // laboratory build the test loads
    static const int fail_rank = [] { const char *e = ab_env("H2_TEST_FAIL_RANK"); return e ? atoi(e) : -1; }();
    if (rc == H2_OK && fail_rank == rank) {
        set_last_error_msg("H2_TEST_FAIL_RANK: injected local failure");
        rc = H2_ERR_HIP;
    }
    return rccl_exchange_and_sum(curve, rc, form, out_kind, d_out, st, "h2_commit_split_rccl_device");
}
