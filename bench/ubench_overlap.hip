// Does a latency/LDS-bound kernel on one stream co-run with the VALU-bound accumulate-like kernel on another?
// A: XYZZ mixed adds, N waves/SIMD resident (grid sized to that);  B: 2^15-bin LDS histogram, 256 x 1024-thread blocks.
// hipcc --offload-arch=gfx950 -O3 -I halo2_amd/csrc bench/ubench_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "curve.cuh"
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__host__ __device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256, 4) k_madd(const u32 *tbl, u32 mask, u32 *out, int iters) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    xyzz<FP> acc = xyzz_identity<FP>();
    affine<FP> nxt = aff_load<FP>(tbl + 16 * (size_t)(t & mask));
    for (int i = 0; i < iters; ++i) {
        affine<FP> p = nxt;
        nxt = aff_load<FP>(tbl + 16 * (size_t)(hash(t * 64u + i + 1) & mask));
        xyzz_madd<FP>(acc, p);
    }
    xyzz_store<FP>(out + 32 * (size_t)t, acc);
}
__global__ void __launch_bounds__(1024) k_lds_hist(u32 *tbl, int per_thread) {
    extern __shared__ u32 h[];
    for (int j = threadIdx.x; j < 32768; j += blockDim.x) h[j] = 0;
    __syncthreads();
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) atomicAdd(&h[hash(i * 131u + k) & 32767u], 1u);
    __syncthreads();
    for (int j = threadIdx.x; j < 32768; j += blockDim.x) tbl[(size_t)blockIdx.x * 32768 + j] = h[j];
}
int main() {
    const size_t big = (size_t)1 << 20;
    u32 *tbl, *out, *hist;
    CK(hipMalloc(&tbl, big * 64)); CK(hipMemset(tbl, 1, big * 64));
    CK(hipMalloc(&out, (size_t)262144 * 128)); CK(hipMalloc(&hist, (size_t)256 * 32768 * 4));
    CK(hipFuncSetAttribute((const void *)k_lds_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    for (int waves = 4; waves >= 2; --waves) {
        int threads = 256 * 4 * waves * 64;  // CUs * SIMDs * waves * 64
        int iters = 64 * 4 / waves;
        float a_ms, b_ms, both_ms;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(k_madd, dim3(threads / 256), dim3(256), 0, s1, tbl, (u32)(big - 1), out, iters); CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&a_ms, e0, e1));
            CK(hipEventRecord(f0, s2)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_lds_hist, dim3(256), dim3(1024), 131072, s2, hist, 64); CK(hipEventRecord(f1, s2));
            CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&b_ms, f0, f1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(k_madd, dim3(threads / 256), dim3(256), 0, s1, tbl, (u32)(big - 1), out, iters); CK(hipEventRecord(e1, s1));
            CK(hipEventRecord(f0, s2)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_lds_hist, dim3(256), dim3(1024), 131072, s2, hist, 64); CK(hipEventRecord(f1, s2));
            CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
            float x, y; CK(hipEventElapsedTime(&x, e0, e1)); CK(hipEventElapsedTime(&y, e0, f1)); both_ms = x > y ? x : y;
        }
        printf("A at %d waves/SIMD: alone %.3f ms; B alone (4 launches) %.3f ms; both streams together %.3f ms (serial would be %.3f)\n", waves, a_ms, b_ms, both_ms, a_ms + b_ms);
    }
    return 0;
}
