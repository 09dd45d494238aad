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

#define ASHR64(A) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(A));
DEF_KERNEL(k_ashrrev_i64, DECL_U64,
           ASHR64(a0) ASHR64(a1) ASHR64(a2) ASHR64(a3) ASHR64(a4) ASHR64(a5) ASHR64(a6) ASHR64(a7), SINK_U64)
#define MADI64(A) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
DEF_KERNEL(k_mad_i64_i32, DECL_U64,
           MADI64(a0) MADI64(a1) MADI64(a2) MADI64(a3) MADI64(a4) MADI64(a5) MADI64(a6) MADI64(a7), SINK_U64)
// the multiplier's column step as it is (mad, mad, mad, bfi, 64-bit shift) and with the shift split into two 32-bit operations:
// eight chains in fixed registers, the step repeated 8 times inside ONE statement (16 + 8 moves / xors of overhead per 320 / 384)
extern "C" __global__ void __launch_bounds__(256) k_col_shift64(u32 *out, int iters, u32 seed) {
    u32 x = seed | 1, y = threadIdx.x * 2654435761u + 12345u, acc = 0, r;
    for (int it = 0; it < iters; ++it) { asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %2\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %1\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %1\n\tv_mov_b32 v51, %2\n\tv_mov_b32 v52, %1\n\tv_mov_b32 v53, %2\n\tv_mov_b32 v54, %1\n\tv_mov_b32 v55, %2\n\t.rept 8\n\tv_mad_i64_i32 v[40:41], vcc, %1, %2, v[40:41]\n\tv_mad_i64_i32 v[40:41], vcc, %2, %1, v[40:41]\n\tv_mad_i64_i32 v[40:41], vcc, %1, %1, v[40:41]\n\tv_bfi_b32 v56, v40, 0, %3\n\tv_ashrrev_i64 v[40:41], 29, v[40:41]\n\tv_mad_i64_i32 v[42:43], vcc, %1, %2, v[42:43]\n\tv_mad_i64_i32 v[42:43], vcc, %2, %1, v[42:43]\n\tv_mad_i64_i32 v[42:43], vcc, %1, %1, v[42:43]\n\tv_bfi_b32 v57, v42, 0, %3\n\tv_ashrrev_i64 v[42:43], 29, v[42:43]\n\tv_mad_i64_i32 v[44:45], vcc, %1, %2, v[44:45]\n\tv_mad_i64_i32 v[44:45], vcc, %2, %1, v[44:45]\n\tv_mad_i64_i32 v[44:45], vcc, %1, %1, v[44:45]\n\tv_bfi_b32 v58, v44, 0, %3\n\tv_ashrrev_i64 v[44:45], 29, v[44:45]\n\tv_mad_i64_i32 v[46:47], vcc, %1, %2, v[46:47]\n\tv_mad_i64_i32 v[46:47], vcc, %2, %1, v[46:47]\n\tv_mad_i64_i32 v[46:47], vcc, %1, %1, v[46:47]\n\tv_bfi_b32 v59, v46, 0, %3\n\tv_ashrrev_i64 v[46:47], 29, v[46:47]\n\tv_mad_i64_i32 v[48:49], vcc, %1, %2, v[48:49]\n\tv_mad_i64_i32 v[48:49], vcc, %2, %1, v[48:49]\n\tv_mad_i64_i32 v[48:49], vcc, %1, %1, v[48:49]\n\tv_bfi_b32 v56, v48, 0, %3\n\tv_ashrrev_i64 v[48:49], 29, v[48:49]\n\tv_mad_i64_i32 v[50:51], vcc, %1, %2, v[50:51]\n\tv_mad_i64_i32 v[50:51], vcc, %2, %1, v[50:51]\n\tv_mad_i64_i32 v[50:51], vcc, %1, %1, v[50:51]\n\tv_bfi_b32 v57, v50, 0, %3\n\tv_ashrrev_i64 v[50:51], 29, v[50:51]\n\tv_mad_i64_i32 v[52:53], vcc, %1, %2, v[52:53]\n\tv_mad_i64_i32 v[52:53], vcc, %2, %1, v[52:53]\n\tv_mad_i64_i32 v[52:53], vcc, %1, %1, v[52:53]\n\tv_bfi_b32 v58, v52, 0, %3\n\tv_ashrrev_i64 v[52:53], 29, v[52:53]\n\tv_mad_i64_i32 v[54:55], vcc, %1, %2, v[54:55]\n\tv_mad_i64_i32 v[54:55], vcc, %2, %1, v[54:55]\n\tv_mad_i64_i32 v[54:55], vcc, %1, %1, v[54:55]\n\tv_bfi_b32 v59, v54, 0, %3\n\tv_ashrrev_i64 v[54:55], 29, v[54:55]\n\t.endr\n\tv_xor_b32 %0, v40, v42\n\tv_xor_b32 %0, %0, v44\n\tv_xor_b32 %0, %0, v46\n\tv_xor_b32 %0, %0, v48\n\tv_xor_b32 %0, %0, v50\n\tv_xor_b32 %0, %0, v52\n\tv_xor_b32 %0, %0, v54\n\tv_xor_b32 %0, %0, v56" : "=&v"(r) : "v"(x), "v"(y), "s"(0x1fffffff) : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59"); acc ^= r; x += acc & 1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
extern "C" __global__ void __launch_bounds__(256) k_col_shift32x2(u32 *out, int iters, u32 seed) {
    u32 x = seed | 1, y = threadIdx.x * 2654435761u + 12345u, acc = 0, r;
    for (int it = 0; it < iters; ++it) { asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %2\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %1\n\tv_mov_b32 v47, %2\n\tv_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %1\n\tv_mov_b32 v51, %2\n\tv_mov_b32 v52, %1\n\tv_mov_b32 v53, %2\n\tv_mov_b32 v54, %1\n\tv_mov_b32 v55, %2\n\t.rept 8\n\tv_mad_i64_i32 v[40:41], vcc, %1, %2, v[40:41]\n\tv_mad_i64_i32 v[40:41], vcc, %2, %1, v[40:41]\n\tv_mad_i64_i32 v[40:41], vcc, %1, %1, v[40:41]\n\tv_bfi_b32 v56, v40, 0, %3\n\tv_alignbit_b32 v40, v41, v40, 29\n\tv_ashrrev_i32 v41, 29, v41\n\tv_mad_i64_i32 v[42:43], vcc, %1, %2, v[42:43]\n\tv_mad_i64_i32 v[42:43], vcc, %2, %1, v[42:43]\n\tv_mad_i64_i32 v[42:43], vcc, %1, %1, v[42:43]\n\tv_bfi_b32 v57, v42, 0, %3\n\tv_alignbit_b32 v42, v43, v42, 29\n\tv_ashrrev_i32 v43, 29, v43\n\tv_mad_i64_i32 v[44:45], vcc, %1, %2, v[44:45]\n\tv_mad_i64_i32 v[44:45], vcc, %2, %1, v[44:45]\n\tv_mad_i64_i32 v[44:45], vcc, %1, %1, v[44:45]\n\tv_bfi_b32 v58, v44, 0, %3\n\tv_alignbit_b32 v44, v45, v44, 29\n\tv_ashrrev_i32 v45, 29, v45\n\tv_mad_i64_i32 v[46:47], vcc, %1, %2, v[46:47]\n\tv_mad_i64_i32 v[46:47], vcc, %2, %1, v[46:47]\n\tv_mad_i64_i32 v[46:47], vcc, %1, %1, v[46:47]\n\tv_bfi_b32 v59, v46, 0, %3\n\tv_alignbit_b32 v46, v47, v46, 29\n\tv_ashrrev_i32 v47, 29, v47\n\tv_mad_i64_i32 v[48:49], vcc, %1, %2, v[48:49]\n\tv_mad_i64_i32 v[48:49], vcc, %2, %1, v[48:49]\n\tv_mad_i64_i32 v[48:49], vcc, %1, %1, v[48:49]\n\tv_bfi_b32 v56, v48, 0, %3\n\tv_alignbit_b32 v48, v49, v48, 29\n\tv_ashrrev_i32 v49, 29, v49\n\tv_mad_i64_i32 v[50:51], vcc, %1, %2, v[50:51]\n\tv_mad_i64_i32 v[50:51], vcc, %2, %1, v[50:51]\n\tv_mad_i64_i32 v[50:51], vcc, %1, %1, v[50:51]\n\tv_bfi_b32 v57, v50, 0, %3\n\tv_alignbit_b32 v50, v51, v50, 29\n\tv_ashrrev_i32 v51, 29, v51\n\tv_mad_i64_i32 v[52:53], vcc, %1, %2, v[52:53]\n\tv_mad_i64_i32 v[52:53], vcc, %2, %1, v[52:53]\n\tv_mad_i64_i32 v[52:53], vcc, %1, %1, v[52:53]\n\tv_bfi_b32 v58, v52, 0, %3\n\tv_alignbit_b32 v52, v53, v52, 29\n\tv_ashrrev_i32 v53, 29, v53\n\tv_mad_i64_i32 v[54:55], vcc, %1, %2, v[54:55]\n\tv_mad_i64_i32 v[54:55], vcc, %2, %1, v[54:55]\n\tv_mad_i64_i32 v[54:55], vcc, %1, %1, v[54:55]\n\tv_bfi_b32 v59, v54, 0, %3\n\tv_alignbit_b32 v54, v55, v54, 29\n\tv_ashrrev_i32 v55, 29, v55\n\t.endr\n\tv_xor_b32 %0, v40, v42\n\tv_xor_b32 %0, %0, v44\n\tv_xor_b32 %0, %0, v46\n\tv_xor_b32 %0, %0, v48\n\tv_xor_b32 %0, %0, v50\n\tv_xor_b32 %0, %0, v52\n\tv_xor_b32 %0, %0, v54\n\tv_xor_b32 %0, %0, v56" : "=&v"(r) : "v"(x), "v"(y), "s"(0x1fffffff) : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59"); acc ^= r; x += acc & 1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

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
#define ALIGNBIT(A) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(A) : "v"(x));
DEF_KERNEL(k_alignbit, DECL_U32, ALIGNBIT(a0) ALIGNBIT(a1) ALIGNBIT(a2) ALIGNBIT(a3) ALIGNBIT(a4) ALIGNBIT(a5) ALIGNBIT(a6) ALIGNBIT(a7), SINK_U32)
#define BFI(A) asm volatile("v_bfi_b32 %0, %0, 0, %1" : "+v"(A) : "v"(x));
DEF_KERNEL(k_bfi, DECL_U32, BFI(a0) BFI(a1) BFI(a2) BFI(a3) BFI(a4) BFI(a5) BFI(a6) BFI(a7), SINK_U32)
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
        {"v_mad_i64_i32", k_mad_i64_i32, 8}, {"v_ashrrev_i64", k_ashrrev_i64, 8},   {"v_alignbit_b32", k_alignbit, 8}, {"v_bfi_b32", k_bfi, 8},
        {"column x8: 3 mad+bfi+shift64", k_col_shift64, 40}, {"column x8: 3 mad+bfi+2 shift32", k_col_shift32x2, 48},
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
