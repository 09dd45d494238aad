// Micro-benchmark: random 4-byte scatter into a 64 MiB array (sorted-index write of the MSM bin pass)
// and random 64-byte gathers from a 64 MiB array (affine base fetch of the bucket accumulation).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
__device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_scatter(u32 *dst, u32 mask, int per) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per; ++k) dst[hash(i * 16u + k) & mask] = i;
}
// scatter with locality: each block writes pairs of adjacent words at random places
__global__ void k_scatter_seq(u32 *dst, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (u32 j = i; j < n; j += gridDim.x * blockDim.x) dst[j] = i;
}
__global__ void k_gather64(const uint4 *src, u32 *out, u32 mask, int per) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < per; ++k) {
        u32 idx = hash(i * 16u + k) & mask;
        const uint4 *p = src + 4 * (size_t)idx;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.x + c.w + d.z;
    }
    out[i] = acc.x ^ acc.y;
}
int main() {
    const u32 n = 1u << 24;
    u32 *dst, *out; uint4 *src;
    CK(hipMalloc(&dst, (size_t)n * 4)); CK(hipMalloc(&src, (size_t)(1 << 20) * 64)); CK(hipMalloc(&out, (1 << 20) * 4));
    CK(hipMemset(src, 1, (size_t)(1 << 20) * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_scatter, dim3((1 << 20) / 256), dim3(256), 0, 0, dst, n - 1, 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("random 4B scatter 16.8M into 64MiB: %.3f ms  %.2f G/s\n", ms, 16.777216 / ms);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_scatter_seq, dim3(2048), dim3(256), 0, 0, dst, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("sequential 4B store 16.8M          : %.3f ms\n", ms);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather64, dim3((1 << 20) / 256), dim3(256), 0, 0, src, out, (1u << 20) - 1, 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("random 64B gather 16.8M from 64MiB : %.3f ms  %.2f G/s  %.1f GB/s\n", ms, 16.777216 / ms, 16.777216 * 64 / ms);
    }
    return 0;
}
