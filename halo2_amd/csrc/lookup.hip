// The data-dependent part of the lookup argument's prover: `permute_expression_pair`
// (halo2_proofs/src/plonk/lookup/prover.rs:557-647).  The reference sorts the input column (`Vec::sort` on field elements,
// ordered by their canonical value), walks it against a BTreeMap of the table column, and hands the leftover table values
// to the rows whose input value repeats the row above.  Here:
//   * a bitonic network sorts both columns (canonical 256-bit keys, eight u32 limb planes in LDS: 2048 keys per workgroup;
//     strides beyond a tile run as global compare-exchange passes, two strides per pass);
//   * "first occurrence" / "run head" flags, two binary searches per row (is this input value in the table? is this table
//     run's value in the input?), two exclusive scans and one gather rebuild the reference's sequential walk:
//       S'[i] = A[i]                      where A[i] differs from A[i-1]                       (:597-607)
//       S'[R[t]] = leftover[r - 1 - t]    for the t-th repeated row R[t]; leftover ascending   (:617-622, `pop()` takes from the end)
//     leftover = the sorted table minus one instance of every distinct input value.
// All integer / comparison work; bound by the global compare-exchange passes (HBM).
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.h"
#include "field.cuh"
#include "host_field.h"

namespace h2 {

namespace {

constexpr int kTileLog = 11;                 // 2048 keys x 32 B = 64 KiB of LDS per workgroup
constexpr u32 kTile = 1u << kTileLog;
constexpr u32 kSortThreads = kTile / 2;      // one compare-exchange per lane per stage

struct key256 {
    u32 v[8];
};

__device__ __forceinline__ bool key_less(const key256 &a, const key256 &b) {
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
    }
    return false;
}

__device__ __forceinline__ bool key_eq(const key256 &a, const key256 &b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

__device__ __forceinline__ key256 key_load(const u32 *p) {
    const uint4 lo = reinterpret_cast<const uint4 *>(p)[0], hi = reinterpret_cast<const uint4 *>(p)[1];
    return key256{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}

__device__ __forceinline__ void key_store(u32 *p, const key256 &k) {
    reinterpret_cast<uint4 *>(p)[0] = make_uint4(k.v[0], k.v[1], k.v[2], k.v[3]);
    reinterpret_cast<uint4 *>(p)[1] = make_uint4(k.v[4], k.v[5], k.v[6], k.v[7]);
}

// copy n elements into the padded sort buffer as canonical keys; the padding sorts last (all ones > any field element)
template <int F>
__global__ void __launch_bounds__(256) lk_prepare(const u32 *__restrict__ src, size_t n, size_t padded, int from_mont, u32 *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded) return;
    if (i < n) {
        fe v = fe_load(src + 8 * i);
        if (from_mont) v = fe_from_mont<F>(v);
        fe_store(dst + 8 * i, v);
    } else {
        key_store(dst + 8 * i, key256{{~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    }
}

template <int F> __global__ void __launch_bounds__(256) lk_finish(const u32 *__restrict__ src, size_t n, int to_mont, u32 *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe v = fe_load(src + 8 * i);
    if (to_mont) v = fe_to_mont<F>(v);
    fe_store(dst + 8 * i, v);
}

// Bitonic stages inside one tile.  tail_k_log == 0: every merge size that fits the tile, completely (j from k/2 down to 1);
// tail_k_log != 0: only the j < kTile part of merge size 2^tail_k_log (its larger strides were global passes).
__global__ void __launch_bounds__(kSortThreads) lk_sort_tile(u32 *__restrict__ a, u32 log_n, u32 tail_k_log) {
    __shared__ u32 plane[8][kTile];
    const u32 tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x << kTileLog;
    const u32 tile_n = log_n < (u32)kTileLog ? (1u << log_n) : kTile;
    for (u32 e = tid; e < tile_n; e += kSortThreads) {
        const key256 k = key_load(a + 8 * (base + e));
#pragma unroll
        for (int l = 0; l < 8; ++l) plane[l][e] = k.v[l];
    }
    __syncthreads();
    const u32 k_first = tail_k_log ? tail_k_log : 1, k_last = tail_k_log ? tail_k_log : (log_n < (u32)kTileLog ? log_n : (u32)kTileLog);
    for (u32 kl = k_first; kl <= k_last; ++kl) {
        const u32 j_top = (kl - 1 < (u32)kTileLog - 1) ? kl - 1 : (u32)kTileLog - 1;
        for (int jl = (int)j_top; jl >= 0; --jl) {
            const u32 j = 1u << jl;
            if (tid < tile_n / 2) {
                const u32 i = ((tid >> jl) << (jl + 1)) | (tid & (j - 1)), l = i | j;
                const bool asc = (((base + i) >> kl) & 1) == 0;
                key256 x, y;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x.v[q] = plane[q][i];
                    y.v[q] = plane[q][l];
                }
                if (key_less(y, x) == asc) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        plane[q][i] = y.v[q];
                        plane[q][l] = x.v[q];
                    }
                }
            }
            __syncthreads();
        }
    }
    for (u32 e = tid; e < tile_n; e += kSortThreads) {
        key256 k;
#pragma unroll
        for (int l = 0; l < 8; ++l) k.v[l] = plane[l][e];
        key_store(a + 8 * (base + e), k);
    }
}

// one global compare-exchange pass: merge size 2^kl, stride 2^jl >= kTile
__global__ void __launch_bounds__(256) lk_sort_global(u32 *__restrict__ a, size_t pairs, u32 kl, u32 jl) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pairs) return;
    const size_t j = (size_t)1 << jl;
    const size_t i = ((t >> jl) << (jl + 1)) | (t & (j - 1)), l = i | j;
    const bool asc = ((i >> kl) & 1) == 0;
    const key256 x = key_load(a + 8 * i), y = key_load(a + 8 * l);
    if (key_less(y, x) == asc) {
        key_store(a + 8 * i, y);
        key_store(a + 8 * l, x);
    }
}

// two global compare-exchange passes in one: strides 2^jl and 2^(jl-1) (both >= kTile) of merge size 2^kl.  A lane owns the four
// keys that differ in those two index bits, so the second stage needs no further memory round trip.
__global__ void __launch_bounds__(256) lk_sort_global2(u32 *__restrict__ a, size_t quads, u32 kl, u32 jl) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= quads) return;
    const size_t j1 = (size_t)1 << jl, j2 = j1 >> 1;
    const size_t low = t & (j2 - 1), high = t >> (jl - 1);
    const size_t i = (high << (jl + 1)) | low;
    const bool asc = ((i >> kl) & 1) == 0;
    key256 e0 = key_load(a + 8 * i), e1 = key_load(a + 8 * (i | j2)), e2 = key_load(a + 8 * (i | j1)), e3 = key_load(a + 8 * (i | j1 | j2));
    auto cx = [&](key256 &x, key256 &y) {
        if (key_less(y, x) == asc) {
            const key256 tmp = x;
            x = y;
            y = tmp;
        }
    };
    cx(e0, e2);
    cx(e1, e3);
    cx(e0, e1);
    cx(e2, e3);
    key_store(a + 8 * i, e0);
    key_store(a + 8 * (i | j2), e1);
    key_store(a + 8 * (i | j1), e2);
    key_store(a + 8 * (i | j1 | j2), e3);
}

// lower bound of `key` in the ascending array s[0..n): first index whose element is not less than key
__device__ __forceinline__ u32 lower_bound(const u32 *__restrict__ s, u32 n, const key256 &key) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (key_less(key_load(s + 8 * (size_t)mid), key)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// rep[i] = 1 where A[i] repeats A[i-1]; keep[i] = 0 for the head of a table run whose value occurs in A (that instance is the
// one paired with the input's first occurrence), else 1.  *missing != 0 when a distinct input value is absent from the table.
__global__ void __launch_bounds__(256) lk_flags(const u32 *__restrict__ A, const u32 *__restrict__ T, u32 n, u32 *__restrict__ rep,
                                                u32 *__restrict__ keep, u32 *__restrict__ missing) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const key256 a = key_load(A + 8 * (size_t)i);
    const bool first = i == 0 || !key_eq(a, key_load(A + 8 * (size_t)(i - 1)));
    rep[i] = first ? 0u : 1u;
    if (first) {
        const u32 p = lower_bound(T, n, a);
        if (p >= n || !key_eq(key_load(T + 8 * (size_t)p), a)) atomicOr(missing, 1u);
    }
    const key256 t = key_load(T + 8 * (size_t)i);
    const bool head = i == 0 || !key_eq(t, key_load(T + 8 * (size_t)(i - 1)));
    u32 k = 1;
    if (head) {
        const u32 p = lower_bound(A, n, t);
        if (p < n && key_eq(key_load(A + 8 * (size_t)p), t)) k = 0;
    }
    keep[i] = k;
}

// exclusive scan of u32 flags, 3 kernels: per-block scan + totals, scan of totals (one block), offsets added on use
constexpr u32 kScanBlock = 1024;

__global__ void __launch_bounds__(kScanBlock) lk_scan_blocks(const u32 *__restrict__ in, u32 n, u32 *__restrict__ out, u32 *__restrict__ totals) {
    __shared__ u32 sh[kScanBlock];
    const u32 i = blockIdx.x * kScanBlock + threadIdx.x;
    const u32 v = i < n ? in[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 d = 1; d < kScanBlock; d <<= 1) {
        const u32 add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (i < n) out[i] = sh[threadIdx.x] - v;
    if (threadIdx.x == kScanBlock - 1) totals[blockIdx.x] = sh[threadIdx.x];
}

__global__ void __launch_bounds__(kScanBlock) lk_scan_totals(u32 *__restrict__ totals, u32 nb, u32 *__restrict__ grand) {
    __shared__ u32 sh[kScanBlock];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nb; base += kScanBlock) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nb ? totals[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (u32 d = 1; d < kScanBlock; d <<= 1) {
            const u32 add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) totals[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand = carry;
}

// leftover[pos] = T[i] for the kept table entries (ascending, as the reference's BTreeMap iterates)
__global__ void __launch_bounds__(256) lk_compact(const u32 *__restrict__ T, u32 n, const u32 *__restrict__ keep, const u32 *__restrict__ keep_pos,
                                                  const u32 *__restrict__ keep_tot, u32 *__restrict__ leftover) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keep[i]) key_store(leftover + 8 * (size_t)(keep_pos[i] + keep_tot[i / kScanBlock]), key_load(T + 8 * (size_t)i));
}

template <int F>
__global__ void __launch_bounds__(256) lk_assemble(const u32 *__restrict__ A, u32 n, const u32 *__restrict__ rep, const u32 *__restrict__ rep_pos,
                                                   const u32 *__restrict__ rep_tot, const u32 *__restrict__ n_rep, const u32 *__restrict__ leftover,
                                                   int to_mont, u32 *__restrict__ out_input, u32 *__restrict__ out_table) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe a = fe_load(A + 8 * (size_t)i);
    fe s = a;
    if (rep[i]) {
        const u32 t = rep_pos[i] + rep_tot[i / kScanBlock];
        s = fe_load(leftover + 8 * (size_t)(*n_rep - 1 - t));
    }
    if (to_mont) {
        a = fe_to_mont<F>(a);
        s = fe_to_mont<F>(s);
    }
    fe_store(out_input + 8 * (size_t)i, a);
    fe_store(out_table + 8 * (size_t)i, s);
}

struct LookupContext {
    std::mutex mu;
    DevBuf keys_a, keys_t, flags, leftover;
    void release_all() {
        keys_a.release();
        keys_t.release();
        flags.release();
        leftover.release();
    }
};
StreamContexts<LookupContext> g_lookup_ctxs;

int ceil_log2(size_t n) {
    int l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

// ascending bitonic sort of 2^log_n canonical keys in `buf`
int sort_padded(u32 *buf, int log_n, hipStream_t st) {
    const size_t n = (size_t)1 << log_n;
    const unsigned tiles = (unsigned)((n + kTile - 1) >> kTileLog);
    hipLaunchKernelGGL(lk_sort_tile, dim3(tiles ? tiles : 1), dim3(kSortThreads), 0, st, buf, (u32)log_n, 0u);
    for (int kl = kTileLog + 1; kl <= log_n; ++kl) {
        int jl = kl - 1;
        for (; jl - 1 >= kTileLog; jl -= 2)          // strides in pairs: half the passes over HBM
            hipLaunchKernelGGL(lk_sort_global2, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, buf, n / 4, (u32)kl, (u32)jl);
        if (jl >= kTileLog)
            hipLaunchKernelGGL(lk_sort_global, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, buf, n / 2, (u32)kl, (u32)jl);
        hipLaunchKernelGGL(lk_sort_tile, dim3(tiles), dim3(kSortThreads), 0, st, buf, (u32)log_n, (u32)kl);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int prepare_and_sort(int field, const void *d_src, size_t n, int form, DevBuf &keys, hipStream_t st) {
    const int log_n = ceil_log2(n ? n : 1);
    const size_t padded = (size_t)1 << log_n;
    int rc = keys.reserve(padded * 32);
    if (rc != H2_OK) return rc;
    dim3 grid((unsigned)((padded + 255) / 256)), block(256);
    const int from_mont = form == H2_FORM_MONTGOMERY;
    if (field == H2_FP) hipLaunchKernelGGL((lk_prepare<FP>), grid, block, 0, st, (const u32 *)d_src, n, padded, from_mont, keys.as<u32>());
    else hipLaunchKernelGGL((lk_prepare<FQ>), grid, block, 0, st, (const u32 *)d_src, n, padded, from_mont, keys.as<u32>());
    return sort_padded(keys.as<u32>(), log_n, st);
}

}  // namespace

void lookup_release_workspaces() { g_lookup_ctxs.release_current_device(); }

}  // namespace h2

using namespace h2;

extern "C" int h2_sort_device(int field, void *d_a, size_t n, int form, void *stream) {
    if ((field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || (n && !d_a) || n > ((size_t)1 << 30))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (n < 2) return H2_OK;
    hipStream_t st = (hipStream_t)stream;
    LookupContext &cx = g_lookup_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = prepare_and_sort(field, d_a, n, form, cx.keys_a, st)) != H2_OK) return rc;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int to_mont = form == H2_FORM_MONTGOMERY;
    if (field == H2_FP) hipLaunchKernelGGL((lk_finish<FP>), grid, block, 0, st, cx.keys_a.as<u32>(), n, to_mont, (u32 *)d_a);
    else hipLaunchKernelGGL((lk_finish<FQ>), grid, block, 0, st, cx.keys_a.as<u32>(), n, to_mont, (u32 *)d_a);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_permute_expression_pair_device(int field, const void *d_input, const void *d_table, size_t n, int form, void *d_permuted_input,
                                                 void *d_permuted_table, void *stream) {
    if ((field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || n > ((size_t)1 << 30) ||
        (n && (!d_input || !d_table || !d_permuted_input || !d_permuted_table)))
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    if (!n) return H2_OK;
    hipStream_t st = (hipStream_t)stream;
    LookupContext &cx = g_lookup_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = prepare_and_sort(field, d_input, n, form, cx.keys_a, st)) != H2_OK) return rc;
    if ((rc = prepare_and_sort(field, d_table, n, form, cx.keys_t, st)) != H2_OK) return rc;
    const u32 un = (u32)n, nb = (un + kScanBlock - 1) / kScanBlock;
    // flags workspace (u32): rep[n] keep[n] rep_pos[n] keep_pos[n] rep_tot[nb] keep_tot[nb] n_rep n_keep missing
    const size_t words = 4 * (size_t)un + 2 * (size_t)nb + 4;
    if ((rc = cx.flags.reserve(words * 4)) != H2_OK) return rc;
    if ((rc = cx.leftover.reserve((size_t)un * 32)) != H2_OK) return rc;
    u32 *rep = cx.flags.as<u32>(), *keep = rep + un, *rep_pos = keep + un, *keep_pos = rep_pos + un;
    u32 *rep_tot = keep_pos + un, *keep_tot = rep_tot + nb, *n_rep = keep_tot + nb, *n_keep = n_rep + 1, *missing = n_keep + 1;
    H2_HIP(hipMemsetAsync(n_rep, 0, 16, st));
    const u32 *A = cx.keys_a.as<u32>(), *T = cx.keys_t.as<u32>();
    dim3 grid((un + 255) / 256), block(256);
    hipLaunchKernelGGL(lk_flags, grid, block, 0, st, A, T, un, rep, keep, missing);
    hipLaunchKernelGGL(lk_scan_blocks, dim3(nb), dim3(kScanBlock), 0, st, (const u32 *)rep, un, rep_pos, rep_tot);
    hipLaunchKernelGGL(lk_scan_totals, dim3(1), dim3(kScanBlock), 0, st, rep_tot, nb, n_rep);
    hipLaunchKernelGGL(lk_scan_blocks, dim3(nb), dim3(kScanBlock), 0, st, (const u32 *)keep, un, keep_pos, keep_tot);
    hipLaunchKernelGGL(lk_scan_totals, dim3(1), dim3(kScanBlock), 0, st, keep_tot, nb, n_keep);
    hipLaunchKernelGGL(lk_compact, grid, block, 0, st, T, un, (const u32 *)keep, (const u32 *)keep_pos, (const u32 *)keep_tot, cx.leftover.as<u32>());
    const int to_mont = form == H2_FORM_MONTGOMERY;
    if (field == H2_FP)
        hipLaunchKernelGGL((lk_assemble<FP>), grid, block, 0, st, A, un, (const u32 *)rep, (const u32 *)rep_pos, (const u32 *)rep_tot, (const u32 *)n_rep,
                           (const u32 *)cx.leftover.as<u32>(), to_mont, (u32 *)d_permuted_input, (u32 *)d_permuted_table);
    else
        hipLaunchKernelGGL((lk_assemble<FQ>), grid, block, 0, st, A, un, (const u32 *)rep, (const u32 *)rep_pos, (const u32 *)rep_tot, (const u32 *)n_rep,
                           (const u32 *)cx.leftover.as<u32>(), to_mont, (u32 *)d_permuted_input, (u32 *)d_permuted_table);
    H2_HIP(hipGetLastError());
    // the reference returns Error::ConstraintSystemFailure from inside the walk (:609-611): the status has to reach the host
    u32 status[3];
    H2_HIP(hipMemcpyAsync(status, n_rep, 12, hipMemcpyDeviceToHost, st));
    H2_HIP(hipStreamSynchronize(st));
    if (status[2]) return H2_ERR_LOOKUP;
    if (status[0] != status[1]) return H2_ERR_LOOKUP;      // cannot happen when nothing is missing; kept as a cross-check
    return H2_OK;
}
