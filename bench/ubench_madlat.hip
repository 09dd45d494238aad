// Dependent-issue latency of v_mad_i64_i32 on gfx950: one wave per SIMD runs a chain of multiply-adds into ONE 64-bit accumulator,
// then the same number split over two and over four independent accumulators (interleaved).  If the time halves with two chains
// the chain is latency-bound (the case of the quad-lane point operations of the fold: one wave, one product at a time).
// hipcc --offload-arch=gfx950 -O3 bench/ubench_madlat.hip -o build/ubench_madlat && ./build/ubench_madlat
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS> __global__ void __launch_bounds__(64) k(int *out, int a, int b, int reps) {
    long long acc0 = threadIdx.x, acc1 = 1, acc2 = 2, acc3 = 3;
    int x1 = 7;
    for (int r = 0; r < reps; ++r) {
        if (CHAINS == 1) {
            asm volatile(
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0"
                : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
        } else if (CHAINS == 2) {
            asm volatile(
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1\n\t"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1"
                : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b) : "vcc");
        } else if (CHAINS == 9) {      // every multiply-add followed by one plain VALU op: does it ride in the multiply-add's shadow?
            asm volatile(
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\t"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\t"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\t"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1"
                : "+v"(acc0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        } else {
            asm volatile(
                "v_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %1, vcc, %4, %5, %1\n\tv_mad_i64_i32 %2, vcc, %4, %5, %2\n\tv_mad_i64_i32 %3, vcc, %4, %5, %3\n\t"
                "v_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %1, vcc, %4, %5, %1\n\tv_mad_i64_i32 %2, vcc, %4, %5, %2\n\tv_mad_i64_i32 %3, vcc, %4, %5, %3"
                : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (int)(acc0 + acc1 + acc2 + acc3) + x1;
}
template <int CHAINS> static void run(int *d, int waves_per_simd) {
    const int reps = 20000, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, 3, 5, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, 3, 5, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (reps * 8.0);
    printf("%s %d, %d wave(s) per SIMD: %.3f ms, %.2f cycles per multiply-add per wave (at 2.4 GHz)\n", CHAINS == 9 ? "mad + add, chains" : "chains", CHAINS == 9 ? 1 : CHAINS, waves_per_simd, ms, cyc);
}
int main() {
    int *d;
    hipMalloc(&d, 256 * 4 * 8 * 64 * 4);       // up to 8 waves per SIMD
    for (int w : {1, 2, 3, 4}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<9>(d, w); }
    return 0;
}
