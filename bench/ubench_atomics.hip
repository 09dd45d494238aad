// Micro-benchmark: throughput of random-address global atomicAdd (the MSM digit histogram / scatter
// primitive) vs an LDS-histogram formulation.  hipcc --offload-arch=gfx950 -O3 bench/ubench_atomics.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
__device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_atomic_noret(u32 *tbl, u32 mask, int per_thread) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) atomicAdd(&tbl[hash(i * 131u + k) & mask], 1u);
}
__global__ void k_atomic_ret(u32 *tbl, u32 *out, u32 mask, int per_thread) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x, s = 0;
    for (int k = 0; k < per_thread; ++k) s += atomicAdd(&tbl[hash(i * 131u + k) & mask], 1u);
    out[i] = s;
}
// LDS histogram of 2^15 bins per block, then flushed with one global atomic per non-empty bin
__global__ void __launch_bounds__(1024) k_lds_hist(u32 *tbl, int per_thread) {
    extern __shared__ u32 h[];
    for (int j = threadIdx.x; j < 32768; j += blockDim.x) h[j] = 0;
    __syncthreads();
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) atomicAdd(&h[hash(i * 131u + k) & 32767u], 1u);
    __syncthreads();
    u32 *dst = tbl + (blockIdx.x & 15) * 32768;
    for (int j = threadIdx.x; j < 32768; j += blockDim.x) if (h[j]) atomicAdd(&dst[j], h[j]);
}
int main() {
    const u32 table = 1u << 19;  // 16 windows x 2^15 buckets
    u32 *tbl, *out;
    CK(hipMalloc(&tbl, table * 4)); CK(hipMemset(tbl, 0, table * 4));
    const int threads = 1 << 20, per = 16;   // 16.8M updates, like one 2^20 MSM with 16 windows
    CK(hipMalloc(&out, threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_atomic_noret, dim3(threads / 256), dim3(256), 0, 0, tbl, table - 1, per);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("global atomicAdd no-return: %.3f ms  %.2f G/s\n", ms, threads * (double)per / ms / 1e6);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_atomic_ret, dim3(threads / 256), dim3(256), 0, 0, tbl, out, table - 1, per);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("global atomicAdd returning: %.3f ms  %.2f G/s\n", ms, threads * (double)per / ms / 1e6);
        CK(hipFuncSetAttribute((const void *)k_lds_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_lds_hist, dim3(256), dim3(1024), 131072, 0, tbl, threads * per / (256 * 1024));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("LDS 2^15-bin hist + flush : %.3f ms  %.2f G/s\n", ms, threads * (double)per / ms / 1e6);
    }
    return 0;
}
