// Micro-benchmark: XYZZ mixed-add throughput ceiling (the MSM accumulate inner loop) under three feeds:
// (a) operand held in registers, (b) gathered from a small L2-resident table, (c) random 64-B gathers
// from a 1 GiB table (the registered-bases layout).  Prices the accumulate kernel against what the
// multiplier can actually sustain.   hipcc --offload-arch=gfx950 -O3 -I halo2_amd/csrc bench/ubench_madd.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve.cuh"
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__host__ __device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, bool LAZY> __global__ void __launch_bounds__(256) k_madd(const u32 *tbl, u32 mask, u32 *out, int iters) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    xyzz<FP> acc = xyzz_identity<FP>();
    affine<FP> p = aff_load<FP>(tbl + 16 * (size_t)(t & mask));
    affine<FP> nxt = p;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { p.x.v[0] ^= i; }   // stays in registers (result is garbage but the work is the same)
        else {
            p = nxt;
            u32 idx = hash(t * 64u + i + 1) & mask;
            nxt = aff_load<FP>(tbl + 16 * (size_t)idx);
        }
        if (LAZY) xyzz_madd_lazy<FP>(acc, p);     // what msm_accumulate runs: no conditional subtraction after a product
        else xyzz_madd<FP>(acc, p);
    }
    if (LAZY) xyzz_reduce_lazy<FP>(acc);
    xyzz_store<FP>(out + 32 * (size_t)t, acc);
}
int main() {
    const size_t big = (size_t)1 << 24;  // 2^24 points x 64 B = 1 GiB
    u32 *tbl, *out;
    CK(hipMalloc(&tbl, big * 64));
    std::vector<u32> h(16 * 65536);
    for (size_t i = 0; i < h.size(); ++i) h[i] = hash((u32)i) & 0x3fffffffu;
    for (size_t off = 0; off < big; off += 65536) CK(hipMemcpy(tbl + 16 * off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int threads = 262144, iters = 64;
    CK(hipMalloc(&out, (size_t)threads * 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        for (int lazy = 0; lazy < 2; ++lazy)
        for (int mode = 0; mode < 3; ++mode) {
            u32 mask = mode == 1 ? 4095u : (u32)(big - 1);
            CK(hipEventRecord(e0));
            if (mode == 0 && !lazy) hipLaunchKernelGGL((k_madd<0, false>), dim3(threads / 256), dim3(256), 0, 0, tbl, mask, out, iters);
            else if (mode == 0) hipLaunchKernelGGL((k_madd<0, true>), dim3(threads / 256), dim3(256), 0, 0, tbl, mask, out, iters);
            else if (!lazy) hipLaunchKernelGGL((k_madd<1, false>), dim3(threads / 256), dim3(256), 0, 0, tbl, mask, out, iters);
            else hipLaunchKernelGGL((k_madd<1, true>), dim3(threads / 256), dim3(256), 0, 0, tbl, mask, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s mode %d (%s): %.3f ms for %.1f M madds -> %.2f G madd/s\n", lazy ? "lazy     " : "canonical", mode,
                   mode == 0 ? "registers" : mode == 1 ? "4096-entry table" : "1 GiB table random", ms, threads * (double)iters / 1e6,
                   threads * (double)iters / ms / 1e6);
        }
    }
    return 0;
}
