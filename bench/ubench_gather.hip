// What does a random 64-byte gather cost the memory system of an MI355X -- 64 bytes or the 128-byte line?
//
// msm_accumulate gathers one 64-byte table point per bucket entry from a 1 GiB table (csrc/msm.hip); rocprofv3's FETCH_SIZE
// reports exactly the bytes such gathers request (bench/ubench_fetch.hip: cal_gather64, ratio 1.01) while it reports HALF the bytes
// of every streaming pattern, so the counter cannot say whether the other half of the line moves too.  Timing can: the kernels
// below all perform the same NUMBER of random accesses over the same 1 GiB and differ only in the bytes per access and in how
// accesses pair up inside a line.
//   g64      one 64-byte slot per lane, every slot of the table at most once                       (the accumulate's pattern)
//   g64pair  lanes 2i / 2i+1 take the two halves of ONE random 128-byte line                       (same bytes, half the lines)
//   g128     one 128-byte line per lane                                                            (twice the bytes, same lines as g64)
//   g256     two consecutive lines per lane                                                        (four times the bytes)
//   stream   the whole table once, 16 bytes per lane, coalesced                                    (the achievable streaming rate)
// If the line is what moves: t(g64) ~ t(g128), t(g64pair) ~ t(g64) / 2.  If 64-byte sectors move: t(g64) ~ t(g128) / 2 and
// t(g64pair) ~ t(g64).  g256 tells a byte limit (t doubles) from a request-rate / TLB limit (t stays).
//
//   hipcc --offload-arch=gfx950 -O3 bench/ubench_gather.hip -o build/ubench/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32;
#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

// slot = an odd multiplier times the access number, masked: a permutation of the slots, no slot twice
__device__ __forceinline__ u32 perm(u32 t, u32 mask) { return (t * 2654435761u + 0x9e3779b9u) & mask; }

template <int BYTES, bool PAIR>
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ src, u32 *__restrict__ sink, u32 naccess, u32 slot_mask) {
    u32 acc = 0;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < naccess; t += gridDim.x * blockDim.x) {
        const uint4 *p;
        if (PAIR) {
            const u32 line = perm(t >> 1, slot_mask >> 1);              // slot_mask counts 64-byte slots
            p = src + 8 * (size_t)line + 4 * (t & 1u);
        } else {
            p = src + (BYTES / 16) * (size_t)perm(t, slot_mask);
        }
#pragma unroll
        for (int i = 0; i < (PAIR ? 4 : BYTES / 16); ++i) {
            const uint4 v = p[i];
            acc ^= v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) stream(const uint4 *__restrict__ src, u32 *__restrict__ sink, size_t n16) {
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F> static float time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t big = (size_t)1 << 30;             // the table of a 2^20-point column at 17-bit windows is 15 x 2^20 x 64 B = 0.94 GiB
    void *buf;
    u32 *sink;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, big));
    CK(hipDeviceSynchronize());
    const dim3 grid(8192), blk(256);
    const u32 n = 1u << 22;                          // accesses per launch (g256: 2^22 x 256 B = the whole table once)
    const uint4 *src = (const uint4 *)buf;
    const float s = time_ms([&] { hipLaunchKernelGGL(stream, grid, blk, 0, 0, src, sink, big / 16); }, 5);
    const float t64 = time_ms([&] { hipLaunchKernelGGL((gather<64, false>), grid, blk, 0, 0, src, sink, n, (u32)(big / 64 - 1)); }, 5);
    const float t64p = time_ms([&] { hipLaunchKernelGGL((gather<64, true>), grid, blk, 0, 0, src, sink, n, (u32)(big / 64 - 1)); }, 5);
    const float t128 = time_ms([&] { hipLaunchKernelGGL((gather<128, false>), grid, blk, 0, 0, src, sink, n, (u32)(big / 128 - 1)); }, 5);
    const float t256 = time_ms([&] { hipLaunchKernelGGL((gather<256, false>), grid, blk, 0, 0, src, sink, n, (u32)(big / 256 - 1)); }, 5);
    // and with four times as many accesses in flight per launch, as the accumulate has (15.7 M gathers per commit)
    const u32 n4 = 1u << 24;
    const float t64b = time_ms([&] { hipLaunchKernelGGL((gather<64, false>), grid, blk, 0, 0, src, sink, n4, (u32)(big / 64 - 1)); }, 3);
    const float t64pb = time_ms([&] { hipLaunchKernelGGL((gather<64, true>), grid, blk, 0, 0, src, sink, n4, (u32)(big / 64 - 1)); }, 3);
    printf("stream  1 GiB                       %8.3f ms  %7.1f GB/s\n", s, big / (s * 1e-3) / 1e9);
    printf("g64     2^22 x  64 B, own line      %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t64, 64.0 * n / (t64 * 1e-3) / 1e9, n / (t64 * 1e-3) / 1e9);
    printf("g64pair 2^22 x  64 B, lines shared  %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t64p, 64.0 * n / (t64p * 1e-3) / 1e9, n / (t64p * 1e-3) / 1e9);
    printf("g128    2^22 x 128 B                %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t128, 128.0 * n / (t128 * 1e-3) / 1e9, n / (t128 * 1e-3) / 1e9);
    printf("g256    2^22 x 256 B                %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t256, 256.0 * n / (t256 * 1e-3) / 1e9, n / (t256 * 1e-3) / 1e9);
    printf("g64     2^24 x  64 B, own line      %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t64b, 64.0 * n4 / (t64b * 1e-3) / 1e9, n4 / (t64b * 1e-3) / 1e9);
    printf("g64pair 2^24 x  64 B, lines shared  %8.3f ms  %7.1f GB/s requested  %6.2f G access/s\n", t64pb, 64.0 * n4 / (t64pb * 1e-3) / 1e9, n4 / (t64pb * 1e-3) / 1e9);
    printf("ratios: t(g64) / t(g128) = %.2f   t(g64pair) / t(g64) = %.2f   t(g256) / t(g128) = %.2f\n", t64 / t128, t64p / t64, t256 / t128);
    return 0;
}
