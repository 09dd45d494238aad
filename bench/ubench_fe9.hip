// Micro-benchmark + self-check of the carry-free 9 x 29-bit field layer (halo2_amd/csrc/field9.cuh) against the production
// 8 x 32-bit layer (field.cuh): modmul / modsqr throughput and the XYZZ mixed-add loop of msm_accumulate in both layers.
//   hipcc --offload-arch=gfx950 -O3 -I halo2_amd/csrc bench/ubench_fe9.hip -o build/ubench_fe9
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "curve9.cuh"
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__host__ __device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int F> __device__ fe rnd_fe(u32 seed) {   // canonical, < 2^254 < p
    fe a;
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = hash(seed * 8 + i);
    a.v[7] &= 0x3fffffffu;
    return a;
}

// out[t] = 1 if every identity holds for lane t's random inputs
template <int F> __global__ void k_check(u32 *bad) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const fe x = rnd_fe<F>(2 * t + 1), y = rnd_fe<F>(2 * t + 2);          // read as reference-Montgomery values
    u32 fail = 0;
    // pack / unpack round trip
    if (!fe_eq(fe9_pack(fe9_unpack(x)), x)) fail |= 1;
    // bridge: r256 -> M9 -> r256
    const fe9 x9 = fe9_from_r256<F>(x), y9 = fe9_from_r256<F>(y);
    if (!fe_eq(fe9_to_r256<F>(x9), x)) fail |= 2;
    // product
    const fe want = fe_mulx<F>(x, y);
    if (!fe_eq(fe9_to_r256<F>(fe9_mul<F>(x9, y9)), want)) fail |= 4;
    if (!fe_eq(fe9_to_r256<F>(fe9_sqr<F>(x9)), fe_mulx<F>(x, x))) fail |= 8;
    // lazy chains with signed, un-normalised operands: (x - y)^2 * (x + y - 3x) etc.
    const fe9 d = fe9_sub(x9, y9), s = fe9_sub(fe9_add(x9, y9), fe9_add(fe9_dbl(x9), x9));
    const fe dd = fe_sub<F>(x, y), ss = fe_sub<F>(fe_add<F>(x, y), fe_add<F>(fe_dbl<F>(x), x));
    if (!fe_eq(fe9_to_r256<F>(fe9_mul<F>(fe9_sqr<F>(d), s)), fe_mulx<F>(fe_mulx<F>(dd, dd), ss))) fail |= 16;
    if (!fe_eq(fe9_to_r256<F>(fe9_norm(s)), ss)) fail |= 32;
    // zero test
    const fe9 z = fe9_sub(fe9_add(x9, y9), fe9_add(y9, x9));
    if (!fe9_is_zero_mod_p<F>(z) || fe9_is_zero_mod_p<F>(d)) fail |= 64;
    fe9 pz = fe9_p_shl<F>(1);
    if (!fe9_is_zero_mod_p<F>(pz) || !fe9_is_zero_mod_p<F>(fe9_sub(fe9_zero(), fe9_p_shl<F>(3)))) fail |= 128;
    // a long dependent chain of mixed adds in both layers
    xyzz<F> acc = xyzz_identity<F>();
    xyzz9<F> acc9 = xyzz9_identity<F>();
    affine<F> pt;
    for (int i = 0; i < 12; i++) {
        pt.x = rnd_fe<F>(1000003 * t + 2 * i);      // not curve points: the formulas are polynomial identities either way
        pt.y = rnd_fe<F>(1000003 * t + 2 * i + 1);
        if (i == 7) { pt.x = fe_zero(); pt.y = fe_zero(); }      // identity operand
        xyzz_madd<F>(acc, pt);
        xyzz9_madd<F>(acc9, aff9_from_r256<F>(pt));
    }
    const xyzz<F> back = xyzz9_to_r256<F>(acc9);
    // projective equality: X/ZZ, Y/ZZZ
    if (!fe_eq(fe_mulx<F>(back.x, acc.zz), fe_mulx<F>(acc.x, back.zz)) || !fe_eq(fe_mulx<F>(back.y, acc.zzz), fe_mulx<F>(acc.y, back.zzz)))
        fail |= 256;
    // P + P and P + (-P) through the mixed add
    affine<F> g;   // (-1, 2) on both curves, Montgomery form
    g.x = fe_neg<F>(fe_one<F>());
    g.y = fe_dbl<F>(fe_one<F>());
    xyzz<F> a1 = xyzz_identity<F>();
    xyzz9<F> b1 = xyzz9_identity<F>();
    xyzz_madd<F>(a1, g); xyzz_madd<F>(a1, g); xyzz_madd<F>(a1, g);
    const aff9<F> g9 = aff9_from_r256<F>(g);
    xyzz9_madd<F>(b1, g9); xyzz9_madd<F>(b1, g9); xyzz9_madd<F>(b1, g9);
    const xyzz<F> b1r = xyzz9_to_r256<F>(b1);
    if (!fe_eq(fe_mulx<F>(b1r.x, a1.zz), fe_mulx<F>(a1.x, b1r.zz)) || !fe_eq(fe_mulx<F>(b1r.y, a1.zzz), fe_mulx<F>(a1.y, b1r.zzz))) fail |= 512;
    affine<F> ng = g;
    ng.y = fe_neg<F>(g.y);
    xyzz9<F> c1 = xyzz9_identity<F>();
    xyzz9_madd<F>(c1, g9); xyzz9_madd<F>(c1, aff9_from_r256<F>(ng));
    if (!xyzz9_is_identity(c1)) fail |= 1024;
    bad[t] = fail;
}

template <int WHAT> __global__ void __launch_bounds__(256) k_rate(u32 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (WHAT <= 1) {
        fe9 x = fe9_unpack(rnd_fe<FP>(t)), y = fe9_unpack(rnd_fe<FP>(t + 77));
        for (int it = 0; it < iters; ++it) {
            if (WHAT == 0) { x = fe9_mul<FP>(x, y); y = fe9_mul<FP>(y, x); }
            else { x = fe9_sqr<FP>(x); y = fe9_sqr<FP>(y); }
        }
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) acc ^= x.v[i] + 3 * y.v[i];
        out[t] = acc;
    } else {
        fe x = rnd_fe<FP>(t), y = rnd_fe<FP>(t + 77);
        for (int it = 0; it < iters; ++it) { x = fe_mul_lazy<FP>(x, y); y = fe_mul_lazy<FP>(y, x); }
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) acc ^= x.v[i] + 3 * y.v[i];
        out[t] = acc;
    }
}

template <int LAYER, int WPS> __global__ void __launch_bounds__(256, WPS) k_madd(const u32 *tbl, u32 mask, u32 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    affine<FP> nxt = aff_load<FP>(tbl + 16 * (size_t)(hash(t) & mask));
    if (LAYER == 0) {
        xyzz<FP> acc = xyzz_identity<FP>();
        for (int i = 0; i < iters; ++i) {
            const affine<FP> p = nxt;
            nxt = aff_load<FP>(tbl + 16 * (size_t)(hash(t * 64u + i + 1) & mask));
            xyzz_madd_lazy<FP>(acc, p);
        }
        xyzz_reduce_lazy<FP>(acc);
        xyzz_store<FP>(out + 32 * (size_t)t, acc);
    } else {
        xyzz9<FP> acc = xyzz9_identity<FP>();
        for (int i = 0; i < iters; ++i) {
            const affine<FP> p = nxt;
            nxt = aff_load<FP>(tbl + 16 * (size_t)(hash(t * 64u + i + 1) & mask));
            xyzz9_madd<FP>(acc, aff9_unpack<FP>(p));      // table entries are stored in M9 form: repacking only
        }
        xyzz_store<FP>(out + 32 * (size_t)t, xyzz9_to_r256<FP>(acc));
    }
}

int main() {
    u32 *d_bad;
    const int nchk = 1 << 16;
    CK(hipMalloc(&d_bad, nchk * 4));
    for (int f = 0; f < 2; f++) {
        if (f == 0) hipLaunchKernelGGL((k_check<FP>), dim3(nchk / 256), dim3(256), 0, 0, d_bad);
        else hipLaunchKernelGGL((k_check<FQ>), dim3(nchk / 256), dim3(256), 0, 0, d_bad);
        CK(hipDeviceSynchronize());
        std::vector<u32> h(nchk);
        CK(hipMemcpy(h.data(), d_bad, nchk * 4, hipMemcpyDeviceToHost));
        u32 any = 0; int cnt = 0;
        for (u32 v : h) { any |= v; cnt += v != 0; }
        printf("fe9 self-check %s: %s (mask 0x%x, %d of %d lanes)\n", f ? "FQ" : "FP", any ? "FAILED" : "ok", any, cnt, nchk);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32 *d_out;
    CK(hipMalloc(&d_out, (size_t)262144 * 128));
    float ms;
    const char *names[3] = {"fe9_mul (9 x 29, carry-free)", "fe9_sqr", "fe_mul_lazy (8 x 32, production)"};
    for (int wps = 8; wps >= 2; wps /= 2)
        for (int what = 0; what < 3; ++what) {
            const int nb = 256 * wps, iters = 1000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (what == 0) hipLaunchKernelGGL((k_rate<0>), dim3(nb), dim3(256), 0, 0, d_out, iters);
                else if (what == 1) hipLaunchKernelGGL((k_rate<1>), dim3(nb), dim3(256), 0, 0, d_out, iters);
                else hipLaunchKernelGGL((k_rate<2>), dim3(nb), dim3(256), 0, 0, d_out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%-36s waves/SIMD %d: %.3f ms  %.1f G/s\n", names[what], wps, ms, (double)nb * 256 * iters * 2 / ms / 1e6);
        }
    const size_t big = (size_t)1 << 24;
    u32 *tbl;
    CK(hipMalloc(&tbl, big * 64));
    std::vector<u32> h(16 * 65536);
    for (size_t i = 0; i < h.size(); ++i) h[i] = hash((u32)i) & 0x3fffffffu;
    for (size_t off = 0; off < big; off += 65536) CK(hipMemcpy(tbl + 16 * off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 64;
    // the chip's clocks move under sustained load: every configuration is timed as 24 back-to-back launches, twice, interleaved
    for (int rep = 0; rep < 2; ++rep)
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int wps = cfg == 0 ? 4 : cfg == 1 ? 4 : cfg == 2 ? 3 : 2;
            const int threads = 65536 * wps;        // exactly one resident round
            std::vector<float> t;
            for (int l = 0; l < 24; ++l) {
                CK(hipEventRecord(e0));
                if (cfg == 0) hipLaunchKernelGGL((k_madd<0, 4>), dim3(threads / 256), dim3(256), 0, 0, tbl, (u32)(big - 1), d_out, iters);
                else if (cfg == 1) hipLaunchKernelGGL((k_madd<1, 4>), dim3(threads / 256), dim3(256), 0, 0, tbl, (u32)(big - 1), d_out, iters);
                else if (cfg == 2) hipLaunchKernelGGL((k_madd<1, 3>), dim3(threads / 256), dim3(256), 0, 0, tbl, (u32)(big - 1), d_out, iters);
                else hipLaunchKernelGGL((k_madd<1, 2>), dim3(threads / 256), dim3(256), 0, 0, tbl, (u32)(big - 1), d_out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            const double m = threads * (double)iters;
            printf("madd loop, 1 GiB random gathers, %s, %d waves/SIMD: median %.3f ms (min %.3f, max %.3f) for %.1f M madds -> %.2f G madd/s (best %.2f)\n",
                   cfg ? "fe9 (9 x 29)" : "fe  (8 x 32)", wps, t[12], t[0], t[23], m / 1e6, m / t[12] / 1e6, m / t[0] / 1e6);
        }
    return 0;
}
