// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the NTT passes (gfx950): kernels that read or write a KNOWN number
// of bytes exactly once, to be run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass).  The guide's x2 correction
// was calibrated on 16 B / lane streaming reads; the second NTT pass also gathers its single-use stage-major twiddles as 64-byte
// runs (planes a, b: four consecutive 16-byte entries per tile row) and 16-byte runs (plane c: four consecutive 4-byte entries),
// rows 2^10 entries apart -- a pattern nobody calibrated.  Every kernel touches 32 MiB (pattern kernels: 32 MiB of payload spread over
// a larger buffer) so reported KiB / 32768 is the correction factor to divide by.
//   hipcc --offload-arch=gfx950 -O3 bench/ubench_fetch.hip -o build/ubench/ubench_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;

// 16 B per lane, consecutive lanes consecutive: the calibrated pattern
extern "C" __global__ void __launch_bounds__(256) cal_stream16(const uint4 *__restrict__ src, u32 *__restrict__ sink, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 32 B per lane as two 16-byte loads (a field element), consecutive lanes consecutive: the NTT's data loads (T = 4 columns: 128-byte
// runs, rows `row_stride` elements apart)
extern "C" __global__ void __launch_bounds__(256) cal_rows128(const uint4 *__restrict__ src, u32 *__restrict__ sink, size_t rows, size_t row_stride32) {
    // lane = (tile, row, col): a tile is 1024 rows x 4 columns of 32 B, rows row_stride32 = 1024 elements apart, tiles 4 elements
    // apart: element = row * 1024 + tile * 4 + col -- every element of the 2^20-element vector exactly once
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; t < rows * 4; t += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = t >> 12, row = (t >> 2) & 1023, col = t & 3;
        const uint4 *p = src + 2 * (row * row_stride32 + tile * 4 + col);
        const uint4 a = p[0], b = p[1];
        acc ^= a.x ^ a.w ^ b.x ^ b.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// runs of `run` lanes x 16 B (64-byte runs for run = 4), runs `stride16` 16-byte slots apart: twiddle planes a / b
extern "C" __global__ void __launch_bounds__(256) cal_runs16(const uint4 *__restrict__ src, u32 *__restrict__ sink, size_t nruns, u32 run, size_t stride16) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; t < nruns * run; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / run, c = t % run;
        const uint4 v = src[(r & 1023) * stride16 + (r >> 10) * run + c];        // consecutive runs stride16 slots apart, the region read exactly once
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// runs of `run` lanes x 4 B (16-byte runs for run = 4): twiddle plane c
extern "C" __global__ void __launch_bounds__(256) cal_runs4(const u32 *__restrict__ src, u32 *__restrict__ sink, size_t nruns, u32 run, size_t stride4) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; t < nruns * run; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / run, c = t % run;
        acc ^= src[(r & 1023) * stride4 + (r >> 10) * run + c];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// one 64-byte point (four 16-byte loads) per lane at a pseudo-random 64-byte slot of a 1 GiB table: msm_accumulate's gathers.  A slot is
// HALF a 128-byte line and the other half belongs to a point nobody near in time wants: the memory system moves the line.
extern "C" __global__ void __launch_bounds__(256) cal_gather64(const uint4 *__restrict__ src, u32 *__restrict__ sink, size_t ngather, u32 slot_mask) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; t < ngather; t += (size_t)gridDim.x * blockDim.x) {
        const u32 slot = ((u32)t * 2654435761u + 0x9e3779b9u) & slot_mask;          // odd multiplier: a permutation of the slots
        const uint4 *p = src + 4 * (size_t)slot;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 32 B per lane written as two 16-byte stores, 4-lane rows `row_stride32` apart (the NTT's second-pass stores); and plain streaming
extern "C" __global__ void __launch_bounds__(256) cal_write_rows128(uint4 *__restrict__ dst, size_t rows, size_t row_stride32) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t < rows * 4; t += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = t >> 12, row = (t >> 2) & 1023, col = t & 3;
        uint4 *p = dst + 2 * (row * row_stride32 + tile * 4 + col);
        p[0] = make_uint4((u32)t, 1, 2, 3);
        p[1] = make_uint4(4, 5, 6, (u32)t);
    }
}
extern "C" __global__ void __launch_bounds__(256) cal_write_stream16(uint4 *__restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((u32)i, 1, 2, 3);
}

int main() {
    const size_t payload = (size_t)32 << 20;            // bytes every kernel moves
    const size_t big = (size_t)1 << 30;                 // the pattern kernels spread their payload over 1 GiB
    void *buf;
    u32 *sink;
    CK(hipMalloc(&buf, big));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, big));
    CK(hipDeviceSynchronize());
    const dim3 grid(4096), blk(256);
    for (int rep = 0; rep < 3; ++rep) {
        // flush the caches between kernels by touching another region is NOT done: every kernel reads a region nobody touched before
        const size_t off = (size_t)rep * ((size_t)256 << 20);     // three disjoint 256 MiB regions... patterns use their own strides
        (void)off;
        hipLaunchKernelGGL(cal_stream16, grid, blk, 0, 0, (const uint4 *)((char *)buf + ((size_t)rep << 25)), sink, payload / 16);
        // 2^20 elements as 2^18 rows of 4 columns, rows 2^10 elements apart inside tiles of 2^20: exactly the second pass's data loads
        hipLaunchKernelGGL(cal_rows128, grid, blk, 0, 0, (const uint4 *)((char *)buf + ((size_t)256 << 20) + ((size_t)rep << 25)), sink, payload / 128, (size_t)1024);
        // 64-byte runs, 16 KiB apart (stage-major twiddle planes a / b: 4 consecutive entries, rows 2^10 entries apart)
        // (16 MiB: one plane of the 2^20-entry table, every entry once)
        hipLaunchKernelGGL(cal_runs16, grid, blk, 0, 0, (const uint4 *)((char *)buf + ((size_t)128 << 20) + ((size_t)rep << 24)), sink, (size_t)262144, 4u, (size_t)1024);
        // 16-byte runs, 4 KiB apart (plane c: 4 MiB)
        hipLaunchKernelGGL(cal_runs4, grid, blk, 0, 0, (const u32 *)((char *)buf + ((size_t)640 << 20) + ((size_t)rep << 22)), sink, (size_t)262144, 4u, (size_t)1024);
        // 2^22 gathers of 64 B (256 MiB requested) over the whole 1 GiB buffer, every slot at most once
        hipLaunchKernelGGL(cal_gather64, grid, blk, 0, 0, (const uint4 *)buf, sink, (size_t)1 << 22, (u32)((1u << 24) - 1));
        hipLaunchKernelGGL(cal_write_stream16, grid, blk, 0, 0, (uint4 *)((char *)buf + ((size_t)768 << 20)), payload / 16);
        hipLaunchKernelGGL(cal_write_rows128, grid, blk, 0, 0, (uint4 *)((char *)buf + ((size_t)832 << 20)), payload / 128, (size_t)1024);
        CK(hipDeviceSynchronize());
    }
    printf("calibration kernels done: cal_stream16 / cal_rows128 / cal_write_* moved %zu bytes each, cal_runs16 16 MiB, cal_runs4 4 MiB\n", payload);
    return 0;
}
