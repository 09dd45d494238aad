// Native parity + throughput check of the gfx950 field arithmetic against the C oracle.
// TEST CODE: links oracle/h2_oracle.c (allowed: tests may use the oracle as the checker).
// Build: hipcc --offload-arch=gfx950 -O3 tests/native/field_check.hip oracle/h2_oracle.c -o build/field_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#define H2_FIELD_EXPERIMENTS 1
#include "../../halo2_amd/csrc/field.cuh"
#include "../../halo2_amd/csrc/curve9.cuh"
#include "../../halo2_amd/csrc/curve9_wide.cuh"

extern "C" {
void orc_f_mul(int field, uint64_t *r, const uint64_t *a, const uint64_t *b);
void orc_f_add(int field, uint64_t *r, const uint64_t *a, const uint64_t *b);
void orc_f_sub(int field, uint64_t *r, const uint64_t *a, const uint64_t *b);
void orc_f_inv(int field, uint64_t *r, const uint64_t *a);
void orc_random_field(int field, uint64_t seed, uint64_t *out, size_t n);
void orc_from_mont(int field, uint64_t *a, size_t n);
}
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// op: 0 mul_c, 1 mul_col, 2 mul(per-product asm), 3 add, 4 sub, 5 inv, 6 mul_blk
template <int F> __global__ void k_ops(const u32 *a, const u32 *b, u32 *out, int n, int op) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(a + 8 * i), y = fe_load(b + 8 * i), r;
    switch (op) {
        case 0: r = fe_mul_c<F>(x, y); break;
        case 1: r = fe_mul_col<F>(x, y); break;
        case 2: r = fe_mul<F>(x, y); break;
        case 3: r = fe_add<F>(x, y); break;
        case 4: r = fe_sub<F>(x, y); break;
        case 6: r = fe_mul_blk<F>(x, y); break;
        case 7: r = fe_mul_sched<F>(x, y); break;
        default: r = fe_inv<F>(x); break;
    }
    fe_store(out + 8 * i, r);
}

// ---- lazy arithmetic (field.cuh "lazy reduction"): every result, made canonical, must equal the canonical computation.
// a, b arrive canonical; `ka`, `kb` in {0, 1, 2} pick the representative a + ka p / b + kb p (skipped when it would not be
// a legal lazy value, i.e. >= 2p + 2^200).  mode 0: mul, 1: sub, 2: is-zero of (a - b), 3: 300 dependent squarings + subs.
template <int F> __device__ fe add_kp(fe v, int k) {
    for (int r = 0; r < k; ++r) {
        u32 c = 0;
        for (int i = 0; i < 8; i++) { u32 co; v.v[i] = __builtin_addc(v.v[i], mod_limb<F>(i), c, &co); c = co; }
    }
    return v;
}
template <int F> __global__ void k_lazy(const u32 *a, const u32 *b, u32 *out, int n, int mode, int ka, int kb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(a + 8 * i), y = fe_load(b + 8 * i);
    // a value may only sit in [2p, 2p + d) if its residue is tiny: keep the representative legal
    const bool x_small = (x.v[7] | x.v[6] | x.v[5] | x.v[4]) == 0, y_small = (y.v[7] | y.v[6] | y.v[5] | y.v[4]) == 0;
    fe xl = add_kp<F>(x, (ka == 2 && !x_small) ? 1 : ka), yl = add_kp<F>(y, (kb == 2 && !y_small) ? 1 : kb);
    fe r;
    if (mode == 0) r = fe_reduce_lazy<F>(fe_mul_lazy<F>(xl, yl));
    else if (mode == 1) r = fe_reduce_lazy<F>(fe_sub_lazy<F>(xl, yl));
    else if (mode == 2) { r = fe_zero(); r.v[0] = fe_is_zero_lazy<F>(fe_sub_lazy<F>(xl, yl)) ? 1u : 0u; }
    else {
        fe u = xl, v = x;
        for (int it = 0; it < 300; ++it) {
            u = fe_sub_lazy<F>(fe_mul_lazy<F>(u, u), yl);      // u <- u^2 - y, lazily
            v = fe_sub<F>(fe_mulx<F>(v, v), y);                // canonically
        }
        r = fe_reduce_lazy<F>(u);
        if (!fe_eq(r, v)) r.v[0] ^= 0xdeadbeefu;               // flagged below as a mismatch against v
        else r = fe_zero();
    }
    fe_store(out + 8 * i, r);
}

template <int F, int IMPL> __global__ void __launch_bounds__(256) k_chain(const u32 *a, u32 *out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    fe x = fe_load(a + 8 * (i & 1023)), y = fe_load(a + 8 * ((i + 7) & 1023));
    fe z = y, w = x;
    for (int it = 0; it < iters; ++it) {
        if (IMPL == 0) { x = fe_mul_c<F>(x, y); z = fe_mul_c<F>(z, w); }
        if (IMPL == 1) { x = fe_mul_col<F>(x, y); z = fe_mul_col<F>(z, w); }
        if (IMPL == 2) { x = fe_mul<F>(x, y); z = fe_mul<F>(z, w); }
        if (IMPL == 3) { x = fe_mul_blk<F>(x, y); z = fe_mul_blk<F>(z, w); }
        if (IMPL == 4) { x = fe_mul_sched<F>(x, y); z = fe_mul_sched<F>(z, w); }
    }
    fe_store(out + 8 * i, fe_add<F>(x, z));
}

// ---- the carry-free 9 x 29 layer (field9.cuh) against the C oracle.  Inputs / outputs cross in the reference's Montgomery form.
// op 0: mul   1: sqr   2: (a - b)^2 (a + b - 3a) on signed un-normalised limbs   3: a - b after a carry pass
// op 4: bridge round trip r256 -> M9 -> r256   5: plain-C multiplier (fe9_mul_c)   6: M9 table form (aff_to_m9 + unpack) times b
// op 8: fe9_dot2 (two products, one reduction)   9: fe9_sqr_minus (subtrahend in the upper columns of the square)
template <int F> __global__ void k_ops9(const u32 *a, const u32 *b, u32 *out, int n, int op) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe x = fe_load(a + 8 * i), y = fe_load(b + 8 * i);
    const fe9 x9 = fe9_from_r256<F>(x), y9 = fe9_from_r256<F>(y);
    fe r;
    switch (op) {
        case 0: r = fe9_to_r256<F>(fe9_mul<F>(x9, y9)); break;
        case 1: r = fe9_to_r256<F>(fe9_sqr<F>(x9)); break;
        case 2: r = fe9_to_r256<F>(fe9_mul<F>(fe9_sqr<F>(fe9_sub(x9, y9)), fe9_sub(fe9_add(x9, y9), fe9_add(fe9_dbl(x9), x9)))); break;
        case 3: r = fe9_to_r256<F>(fe9_norm(fe9_sub(x9, y9))); break;
        case 4: r = fe9_to_r256<F>(x9); break;
        case 5: r = fe9_to_r256<F>(fe9_mul_c<F>(x9, y9)); break;
        case 7: r = fe_redc<F>(x); break;                                  // = fe_from_mont: the sort kernels' canonicalisation
        case 8: r = fe9_to_r256<F>(fe9_dot2<F>(x9, y9, fe9_sub(x9, y9), fe9_sub(fe9_zero(), fe9_add(x9, y9)))); break;   // a b - (a - b)(a + b), signed limbs
        case 9: r = fe9_to_r256<F>(fe9_sqr_minus<F>(fe9_sub(x9, y9), fe9_add(fe9_dbl(x9), y9))); break;                   // (a - b)^2 - (2 a + b)
        default: {
            affine<F> pt{x, x};
            const aff9<F> q = aff9_unpack<F>(aff_to_m9<F>(pt));
            r = fe9_to_r256<F>(fe9_mul<F>(q.x, y9));
        }
    }
    fe_store(out + 8 * i, r);
}
// ---- quad-lane point doubling on the carry-free layer (curve9_wide.cuh) against the 8 x 32 one (curve_wide.cuh) ---------------
// mode 0: a chain of `reps` doublings from the affine point (a[i], b[i]) (the formulas are polynomial identities: the point need
// not lie on the curve).  mode 1: ONE doubling from a raw M9 state -- the recorded state whose Y^2 leaves the multiplier with
// limb 0 equal to 2^29 (the only limb value 4 * limb does not fit an i32 for; fe9_quadruple_norm).
template <int F> __global__ void k_dbl9_wide(const u32 *a, const u32 *b, const u32 *raw, u32 *out, int n, int reps, int mode) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (i >= n) return;
    xyzz<F> r;
    xyzz9<F> r9;
    if (mode == 0) {
        const affine<F> p{fe_load(a + 8 * i), fe_load(b + 8 * i)};
        r = xyzz_identity<F>();
        xyzz_madd<F>(r, p);
        r9 = xyzz9_from_r256_wide<F>(r);
    } else {
        r9 = xyzz9_load_raw<F>(raw);
        r = xyzz9_to_r256_wide<F>(r9);
    }
    for (int k = 0; k < reps; ++k) {
        r = xyzz_dbl_wide<F>(r);
        r9 = xyzz9_dbl_wide<F>(r9);
    }
    const xyzz<F> o = xyzz9_to_r256_wide<F>(r9);
    bool same = true;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        same = same && o.x.v[j] == r.x.v[j] && o.y.v[j] == r.y.v[j] && o.zz.v[j] == r.zz.v[j] && o.zzz.v[j] == r.zzz.v[j];
    if ((threadIdx.x & (kGroup - 1)) == 0) out[i] = same ? 0u : 1u;
}
__global__ void k_quadruple_norm(u32 *out) {
    fe9 a = fe9_zero();
    a.v[0] = 1 << 29;                                      // what a product may leave in limb 0
    for (int i = 1; i < 8; i++) a.v[i] = (i32)M29 - i;
    a.v[8] = -5;
    const fe9 got = fe9_quadruple_norm(a), want = fe9_norm(fe9_dbl(fe9_norm(fe9_dbl(fe9_norm(a)))));
    u32 bad = 0;
    for (int i = 0; i < 9; i++) bad |= (u32)(got.v[i] != want.v[i]);
    out[0] = bad;
}
template <int F> int run_wide9(const u32 *da, const u32 *db, int n) {
    static const u32 kState[36] = {     // Fq, M9 limbs of (X, Y, ZZ, ZZZ): doubling 221 of a 16-bit table chain, found on the device
        0x038d27b2, 0x05039a9d, 0x12356f7c, 0x15b60e05, 0x0faf35a4, 0x07a6a677, 0x08780ea0, 0x03734c88, 0x00079790,
        0x01198606, 0x150a0983, 0x15963169, 0x0228e707, 0x076b97cb, 0x15217f1f, 0x0be3beab, 0x1454604c, 0xffe0f2ac,
        0x0413c791, 0x16925096, 0x1558a0c2, 0x09ab6c99, 0x1f698236, 0x043fa20f, 0x1804ede6, 0x0404f056, 0x00394ee7,
        0x00ae9230, 0x090ab3d3, 0x0c3cf140, 0x1de44147, 0x0ac239cf, 0x12ae76e2, 0x0bd3b3ec, 0x1aa2ffd3, 0x00369c3d};
    u32 *draw, *dflag;
    CK(hipMalloc(&draw, sizeof(kState)));
    CK(hipMalloc(&dflag, 4 * (size_t)n));
    CK(hipMemcpy(draw, kState, sizeof(kState), hipMemcpyHostToDevice));
    std::vector<u32> flag(n);
    int fails = 0;
    for (int mode = 0; mode < (F == FQ ? 2 : 1); ++mode) {
        const int cnt = mode ? 1 : n, reps = mode ? 1 : 48;
        hipLaunchKernelGGL((k_dbl9_wide<F>), dim3((cnt * kGroup + 255) / 256), dim3(256), 0, 0, da, db, draw, dflag, cnt, reps, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(flag.data(), dflag, 4 * (size_t)cnt, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < cnt; ++i) bad += flag[i] != 0;
        printf("field %d %-16s: %d/%d mismatches\n", F, mode ? "dbl9 wide 2^29" : "dbl9 wide chain", bad, cnt);
        fails += bad != 0;
    }
    hipLaunchKernelGGL(k_quadruple_norm, dim3(1), dim3(1), 0, 0, dflag);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(flag.data(), dflag, 4, hipMemcpyDeviceToHost));
    printf("field %d %-16s: %d/1 mismatches\n", F, "quadruple norm", (int)flag[0]);
    fails += flag[0] != 0;
    (void)hipFree(draw);
    (void)hipFree(dflag);
    return fails;
}
template <int F> int run_field9(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, const u32 *da, const u32 *db, u32 *dout, int n) {
    std::vector<uint64_t> got(4 * (size_t)n);
    const char *names[] = {"fe9 mul", "fe9 sqr", "fe9 signed chain", "fe9 sub+norm", "fe9 bridge", "fe9 mul_c", "fe9 table form", "fe_redc", "fe9 dot2", "fe9 sqr_minus"};
    int fails = 0;
    for (int op = 0; op < 10; ++op) {
        hipLaunchKernelGGL((k_ops9<F>), dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n, op);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, 32 * (size_t)n, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            uint64_t w[4], t[4], u[4];
            const uint64_t *x = &a[4 * i], *y = &b[4 * i];
            if (op == 0 || op == 5 || op == 6) orc_f_mul(F, w, x, y);
            else if (op == 1) orc_f_mul(F, w, x, x);
            else if (op == 2) {
                orc_f_sub(F, t, x, y); orc_f_mul(F, t, t, t);                    // (a - b)^2
                orc_f_add(F, u, x, y); orc_f_sub(F, u, u, x); orc_f_sub(F, u, u, x); orc_f_sub(F, u, u, x);
                orc_f_mul(F, w, t, u);
            } else if (op == 3) orc_f_sub(F, w, x, y);
            else if (op == 7) { memcpy(w, x, 32); orc_from_mont(F, w, 1); }
            else if (op == 8) {
                orc_f_mul(F, w, x, y); orc_f_sub(F, t, x, y); orc_f_add(F, u, x, y); orc_f_mul(F, t, t, u); orc_f_sub(F, w, w, t);
            } else if (op == 9) {
                orc_f_sub(F, t, x, y); orc_f_mul(F, t, t, t); orc_f_add(F, u, x, x); orc_f_add(F, u, u, y); orc_f_sub(F, w, t, u);
            } else memcpy(w, x, 32);
            if (memcmp(w, &got[4 * i], 32)) { if (!bad) printf("  first mismatch %s idx %d\n", names[op], i); bad++; }
        }
        printf("field %d %-16s: %d/%d mismatches\n", F, names[op], bad, n);
        fails += bad;
    }
    return fails;
}

template <int F> int run_field() {
    const int n = 1 << 14;
    std::vector<uint64_t> a(4 * n), b(4 * n), want(4 * n), got(4 * n);
    orc_random_field(F, 11 + F, a.data(), n);
    orc_random_field(F, 23 + F, b.data(), n);
    // edge cases: 0, 1 (canonical one, not Montgomery), p-1, equal operands
    const uint64_t P[2][4] = {{0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL},
                              {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL}};
    memset(&a[0], 0, 32);
    memset(&a[4], 0, 32); a[4] = 1;
    memcpy(&a[8], P[F], 32); a[8] -= 1;
    memcpy(&b[8], P[F], 32); b[8] -= 1;
    memcpy(&a[12], &b[12], 32);
    memcpy(&b[16], P[F], 32); b[16] -= 1; memset(&a[16], 0, 32); a[16] = 1;
    u32 *da, *db, *dout;
    CK(hipMalloc(&da, 32 * n)); CK(hipMalloc(&db, 32 * n)); CK(hipMalloc(&dout, 32 * n));
    CK(hipMemcpy(da, a.data(), 32 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 32 * n, hipMemcpyHostToDevice));
    const char *names[] = {"mul_c", "mul_col", "mul_asm", "add", "sub", "inv", "mul_blk", "mul_sched"};
    int fails = 0;
    for (int op = 0; op < 8; ++op) {
        int cnt = op == 5 ? 256 : n;
        hipLaunchKernelGGL((k_ops<F>), dim3((cnt + 255) / 256), dim3(256), 0, 0, da, db, dout, cnt, op);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, 32 * cnt, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < cnt; ++i) {
            uint64_t w[4];
            if (op <= 2 || op >= 6) orc_f_mul(F, w, &a[4 * i], &b[4 * i]);
            else if (op == 3) orc_f_add(F, w, &a[4 * i], &b[4 * i]);
            else if (op == 4) orc_f_sub(F, w, &a[4 * i], &b[4 * i]);
            else orc_f_inv(F, w, &a[4 * i]);
            if (memcmp(w, &got[4 * i], 32)) { if (!bad) printf("  first mismatch op %s idx %d\n", names[op], i); bad++; }
        }
        printf("field %d %-8s: %d/%d mismatches\n", F, names[op], bad, cnt);
        fails += bad;
    }
    // lazy arithmetic against the canonical oracle results, every pair of representatives
    for (int mode = 0; mode < 4; ++mode)
        for (int ka = 0; ka < 3; ++ka)
            for (int kb = 0; kb < 3; ++kb) {
                hipLaunchKernelGGL((k_lazy<F>), dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n, mode, ka, kb);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), dout, 32 * n, hipMemcpyDeviceToHost));
                int bad = 0;
                for (int i = 0; i < n; ++i) {
                    uint64_t w[4] = {0, 0, 0, 0};
                    if (mode == 0) orc_f_mul(F, w, &a[4 * i], &b[4 * i]);
                    else if (mode == 1) orc_f_sub(F, w, &a[4 * i], &b[4 * i]);
                    else if (mode == 2) w[0] = memcmp(&a[4 * i], &b[4 * i], 32) == 0;
                    if (memcmp(w, &got[4 * i], 32)) { if (!bad) printf("  first lazy mismatch mode %d reps (%d, %d) idx %d\n", mode, ka, kb, i); bad++; }
                }
                if (bad) printf("field %d lazy mode %d reps (%d, %d): %d/%d mismatches\n", F, mode, ka, kb, bad, n);
                fails += bad;
            }
    printf("field %d lazy mul / sub / zero-test / 300-step chain over 9 representative pairs: done\n", F);
    fails += run_field9<F>(a, b, da, db, dout, n);
    fails += run_wide9<F>(da, db, n);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
    return fails;
}

template <int IMPL> int bench(const char *name, int blocks_per_cu) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * blocks_per_cu;
    std::vector<uint64_t> a(4 * 1024);
    orc_random_field(0, 5, a.data(), 1024);
    u32 *da, *dout;
    CK(hipMalloc(&da, 32 * 1024)); CK(hipMalloc(&dout, (size_t)32 * blocks * 256));
    CK(hipMemcpy(da, a.data(), 32 * 1024, hipMemcpyHostToDevice));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chain<FP, IMPL>), dim3(blocks), dim3(256), 0, 0, da, dout, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<FP, IMPL>), dim3(blocks), dim3(256), 0, 0, da, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)blocks * 256 * iters * 2;
    printf("%-8s waves/SIMD %d: %8.3f ms  %8.2f G modmul/s\n", name, blocks_per_cu, ms, muls / (ms * 1e-3) / 1e9);
    CK(hipFree(da)); CK(hipFree(dout));
    return 0;
}

int main() {
    int fails = run_field<FP>() + run_field<FQ>();
    for (int w : {1, 2, 4, 8}) {
        bench<1>("mul_col", w); bench<3>("mul_blk", w); bench<4>("mul_sched", w);
    }
    printf(fails ? "FIELD CHECK FAILED\n" : "FIELD CHECK OK\n");
    return fails ? 1 : 0;
}
