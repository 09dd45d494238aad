// VALU instruction-rate micro-benchmark for gfx950: which integer/FP64 multiply path should the
// 255-bit Montgomery arithmetic be built on?  (SURVEY.md section 7 "measure before committing to a
// limb layout".)  Standalone: hipcc --offload-arch=gfx950 -O3 bench/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32;
typedef uint64_t u64;

// Each kernel: 8 independent dependency chains per lane, BODY repeated 8x per loop iteration.
#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                            \
    extern "C" __global__ void __launch_bounds__(256) NAME(u32 *out, int iters, u32 seed) { \
        DECL;                                                                         \
        for (int it = 0; it < iters; ++it) {                                          \
            BODY BODY BODY BODY BODY BODY BODY BODY                                   \
        }                                                                             \
        SINK;                                                                         \
    }

#define DECL_U64 u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    u32 x = seed | 1, y = threadIdx.x * 2654435761u + 12345u
#define SINK_U64 out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (u32)((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) >> 32)

#define MAD64(A) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "s10", "s11");
DEF_KERNEL(k_mad_u64_u32, DECL_U64,
           MAD64(a0) MAD64(a1) MAD64(a2) MAD64(a3) MAD64(a4) MAD64(a5) MAD64(a6) MAD64(a7), SINK_U64)

#define LSHLADD(A) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(A) : "v"(a7));
DEF_KERNEL(k_lshl_add_u64, DECL_U64,
           LSHLADD(a0) LSHLADD(a1) LSHLADD(a2) LSHLADD(a3) LSHLADD(a4) LSHLADD(a5) LSHLADD(a6) LSHLADD(a0), SINK_U64)

#define DECL_U32 u32 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    u32 x = seed | 1, y = threadIdx.x * 2654435761u + 12345u
#define SINK_U32 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ x ^ y

#define OP3(OP, A) asm volatile(OP " %0, %0, %1" : "+v"(A) : "v"(x));
#define MULLO(A) OP3("v_mul_lo_u32", A)
DEF_KERNEL(k_mul_lo_u32, DECL_U32, MULLO(a0) MULLO(a1) MULLO(a2) MULLO(a3) MULLO(a4) MULLO(a5) MULLO(a6) MULLO(a7), SINK_U32)
#define MULHI(A) OP3("v_mul_hi_u32", A)
DEF_KERNEL(k_mul_hi_u32, DECL_U32, MULHI(a0) MULHI(a1) MULHI(a2) MULHI(a3) MULHI(a4) MULHI(a5) MULHI(a6) MULHI(a7), SINK_U32)
#define ADD32(A) OP3("v_add_u32", A)
DEF_KERNEL(k_add_u32, DECL_U32, ADD32(a0) ADD32(a1) ADD32(a2) ADD32(a3) ADD32(a4) ADD32(a5) ADD32(a6) ADD32(a7), SINK_U32)
#define MAD24(A) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
DEF_KERNEL(k_mad_u32_u24, DECL_U32, MAD24(a0) MAD24(a1) MAD24(a2) MAD24(a3) MAD24(a4) MAD24(a5) MAD24(a6) MAD24(a7), SINK_U32)
#define MULHI24(A) OP3("v_mul_hi_u32_u24", A)
DEF_KERNEL(k_mul_hi_u32_u24, DECL_U32, MULHI24(a0) MULHI24(a1) MULHI24(a2) MULHI24(a3) MULHI24(a4) MULHI24(a5) MULHI24(a6) MULHI24(a7), SINK_U32)
// carry pair: add_co + addc_co (one 64-bit add the GCN way) counted as 2 instructions
#define ADDC(A, B) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(A), "+v"(B) : "v"(x), "v"(y) : "vcc");
DEF_KERNEL(k_add_co_addc, DECL_U32, ADDC(a0, a1) ADDC(a2, a3) ADDC(a4, a5) ADDC(a6, a7) ADDC(a0, a1) ADDC(a2, a3) ADDC(a4, a5) ADDC(a6, a7), SINK_U32)
// mad + addc of its carry (the Comba column step): counted as 2 instructions
#define MADC(A, H) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(A), "+v"(H) : "v"(x), "v"(y) : "vcc");
#define DECL_MADC DECL_U64; u32 h0 = 0, h1 = 0, h2 = 0, h3 = 0
#define SINK_MADC SINK_U64; out[blockIdx.x * blockDim.x + threadIdx.x] ^= h0 ^ h1 ^ h2 ^ h3
DEF_KERNEL(k_mad_addc, DECL_MADC, MADC(a0, h0) MADC(a1, h1) MADC(a2, h2) MADC(a3, h3) MADC(a4, h0) MADC(a5, h1) MADC(a6, h2) MADC(a7, h3), SINK_MADC)

#define DECL_F64 double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    double x = 1.0000001, y = 1e-9 * threadIdx.x
#define SINK_F64 out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define FMA64(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
DEF_KERNEL(k_fma_f64, DECL_F64, FMA64(a0) FMA64(a1) FMA64(a2) FMA64(a3) FMA64(a4) FMA64(a5) FMA64(a6) FMA64(a7), SINK_F64)
#define DECL_F32 float a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    float x = 1.0000001f, y = 1e-9f * threadIdx.x
#define FMA32(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
DEF_KERNEL(k_fma_f32, DECL_F32, FMA32(a0) FMA32(a1) FMA32(a2) FMA32(a3) FMA32(a4) FMA32(a5) FMA32(a6) FMA32(a7), SINK_F64)

typedef void (*kern_t)(u32 *, int, u32);
struct Entry { const char *name; kern_t k; int instr_per_body; };

int main(int argc, char **argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clk_ghz = prop.clockRate / 1e6;
    printf("device %s  CUs %d  clock %.3f GHz\n", prop.name, cus, clk_ghz);
    int blocks = cus * waves_per_simd;   // 256-thread blocks = 4 waves = 1 wave per SIMD each
    u32 *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    Entry es[] = {
        {"v_add_u32", k_add_u32, 8},         {"v_mad_u32_u24", k_mad_u32_u24, 8},   {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8},   {"v_mul_hi_u32", k_mul_hi_u32, 8},     {"v_mad_u64_u32", k_mad_u64_u32, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_add_co+v_addc_co", k_add_co_addc, 16}, {"v_mad_u64_u32+v_addc", k_mad_addc, 16},
        {"v_fma_f32", k_fma_f32, 8},         {"v_fma_f64", k_fma_f64, 8},
    };
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-24s %12s %14s %16s\n", "instruction", "ms", "Tinstr/s", "cyc/wave-instr/SIMD");
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 100, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double lane_instr = (double)blocks * 256 * iters * 8.0 * e.instr_per_body;
        double wave_instr_per_simd = (double)waves_per_simd * iters * 8.0 * e.instr_per_body;
        double cyc = ms * 1e-3 * clk_ghz * 1e9 / wave_instr_per_simd;
        printf("%-24s %12.3f %14.3f %16.2f\n", e.name, ms, lane_instr / (ms * 1e-3) / 1e12, cyc);
    }
    CK(hipFree(out));
    return 0;
}
