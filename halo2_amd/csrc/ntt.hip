// Radix-2 NTT over the Pasta fields on gfx950.
//
// Replaces the body of `best_fft` (halo2_proofs/src/arithmetic.rs:192-295) and the EvaluationDomain
// wrappers around it (poly/domain.rs:227-255, :303-325, :357-383).
//
// The reference bit-reverses, builds a twiddle table omega^0..omega^(n/2-1) and runs log n
// decimation-in-time butterfly stages (recursively).  This implementation keeps EXACTLY that butterfly
// network (same pairs, same twiddle omega^(i * n/m) per pair), so outputs agree element for element for
// any omega -- but executes it as ceil(log n / 8) HBM passes of up to 8 stages each:
//
//   * a workgroup stages a tile of 2^r rows x T columns of 32-byte elements in LDS (<= 64 KiB, two
//     workgroups per CU), runs r butterfly stages with one barrier each, one butterfly per lane per stage;
//   * tiles are chosen so that every global access is a run of T*32 contiguous bytes (or 2^r*32 on the
//     transposing store of the first pass); the bit-reversal permutation is folded into the first pass's
//     gather, the 1/n scale (ifft) and the zeta coset factors into the first load / last store;
//   * LDS holds each element as two 16-byte halves in separate planes, so consecutive lanes hit
//     consecutive 16-byte slots (ds_read_b128 conflict-free);
//   * twiddles come from a per-(field, omega, log n) table kept in HBM (n/2 x 32 B, built on the device
//     once and cached; the reference rebuilds it serially on every call, :215-221).  Lanes of a wave
//     read consecutive / broadcast entries; the table is L2/MALL resident.
//
// Cost model: (n/2) log n butterflies = 1 modular multiply + add + sub each; ~1.2e3 VALU cycles per
// wave-butterfly against ~100 cycles of LDS + L2 traffic: VALU-bound, like the MSM.  No MFMA.
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <vector>

#include "common.h"
#include "field9.cuh"
#include "host_field.h"

namespace h2 {

struct feparam {
    u32 v[8];
};
static feparam to_param(const u64 a[4]) {
    feparam p;
    memcpy(p.v, a, 32);
    return p;
}
__device__ __forceinline__ fe from_param(const feparam &p) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p.v[i];
    return r;
}

// ---- twiddle table: tw[e] = omega^e, e < count (Montgomery) --------------------------------------
// thread t owns e = t, t + T, t + 2T, ...: omega^t by square-and-multiply, then repeated * omega^T
template <int F>
__global__ void __launch_bounds__(256) ntt_twiddles(u32 *__restrict__ tw, feparam omega_p, feparam step_p, u32 T, size_t count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    fe omega = from_param(omega_p), step = from_param(step_p);
    fe cur = fe_one<F>();
    for (int b = 31 - __clz(t | 1); b >= 0; --b) {
        cur = fe_sqr<F>(cur);
        if ((t >> b) & 1) cur = fe_mulx<F>(cur, omega);
    }
    for (size_t e = t; e < count; e += T) {
        fe_store(tw + 8 * e, cur);
        cur = fe_mulx<F>(cur, step);
    }
}

__device__ __forceinline__ u32 bitrev(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// Workgroup -> tile map of the passes after the first.  A tile is (hi, lo): elements hi 2^(s0+r) + mid 2^s0 + lo T + col; its
// in-pass twiddles omega^((low 2^s0 + lo T + col) ...) depend on `lo` alone.  So (1) the tiles of one `lo` -- one per `hi` -- read
// the SAME table entries, and (2) neighbouring `lo` read NEIGHBOURING entries: T consecutive entries per tile row, i.e. T / 8 of a
// 128-byte line of the two 16-byte planes and T / 32 of a line of the 4-byte plane.  The dispatcher puts workgroup b on XCD b % 8
// (observed, MI355X_MICROARCH.md -- a speed assumption only: any placement gives the same results), each XCD with its own L2.
// With tile = b the 32 / T tiles that share a line sat on as many different XCDs and each fetched the line for itself: the second
// pass of a 2^20 transform fetched 166 MiB for 68 MiB of data + twiddles (profiles/r04_pmc_traffic.json; FETCH_SIZE calibrated
// on these very patterns, bench/ubench_fetch.hip).  Here XCD x takes the x-th CONTIGUOUS EIGHTH of the `lo` range, for every
// `hi`; in dispatch order `lo` runs fastest (the line sharers run side by side), then `hi` (the next tiles re-read what the XCD's
// L2 already holds): 72 MiB at 2^20.  Measured alternatives: a contiguous eighth of the TILES (right for one `hi`, but at 2^22
// every XCD then needs every twiddle of the middle pass: 159 -> 266 MiB), groups of the 32 / T line sharers dealt round-robin
// (2^20: 81 MiB, the last pass of 2^22 319 against 272).
__device__ __forceinline__ u32 ntt_tile_of_block(u32 b, u32 nblocks, int s0, int logT) {
    const int lt = s0 - logT;                           // log2 tiles per hi
    if (lt < 3 || (nblocks & 7u)) return b;             // fewer than eight tiles per hi: dispatch order as it is
    const u32 xcd = b & 7u, k = b >> 3;                 // k-th workgroup of its XCD
    const u32 lo_local = k & ((1u << (lt - 3)) - 1u), hi = k >> (lt - 3);
    return (hi << lt) | (xcd << (lt - 3)) | lo_local;
}

struct PassArgs {
    int L;        // log2 n
    int s0;       // first stage of this pass (stage t pairs x and x + 2^t)
    int r;        // stages in this pass
    int logT;     // log2 of tile columns
    int first;    // 1: gather input through the bit-reversal permutation, transposing store
    int last;     // 1: results leave the transform: store canonical values (between passes they stay lazy, field.cuh)
    int load_mode;   // 0 none; 1: x {1, k0, k1}[j % 3] for j < n_in, zero for j >= n_in (coeff_to_extended)
    int store_mode;  // 0 none; 1: x k0 (ifft divisor); 2: x {k0, k1, k2}[x % 3] (extended_to_coeff)
    size_t n_in;     // valid input elements (first pass); elements beyond are read as zero
    feparam lk0, lk1;       // load multipliers
    feparam k0, k1, k2;     // store multipliers
};

// LDS planes: lo16[slot], hi16[slot], slot = mid * T + col.
// R = stages in the pass (compile time, so the twiddles of all R stages can sit in registers: their loads
// are issued together with the tile load instead of one L2 round trip per stage).
template <int F, int R, bool FIRST>
__global__ void __launch_bounds__(1024) ntt_pass(const u32 *__restrict__ in, u32 *__restrict__ out,
                                                 const u32 *__restrict__ tw, PassArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int r = R;
    const int logT = A.logT, L = A.L, s0 = A.s0;
    const u32 T = 1u << logT, rows = 1u << r, tile = rows << logT;
    uint4 *lo16 = lds, *hi16 = lds + tile;
    const u32 tid = threadIdx.x, nthr = blockDim.x;

    // tile coordinates
    size_t hi_idx = 0, lo0 = 0;   // general pass: x = hi_idx * 2^(s0+r) + mid * 2^s0 + lo0 + col
    u32 c0 = 0;                   // first pass: columns c0 .. c0 + T - 1 of the 2^(L-r) column space
    if (FIRST) {
        c0 = blockIdx.x << logT;
    } else {
        u32 tiles_per_hi = 1u << (s0 - logT);
        hi_idx = blockIdx.x / tiles_per_hi;
        lo0 = (size_t)(blockIdx.x % tiles_per_hi) << logT;
    }

    // ---- load ----
    if (FIRST) {
        const int cb = L - r;  // column bits
        for (u32 e = tid; e < tile; e += nthr) {
            u32 col = e & (T - 1), row = e >> logT;
            size_t j = ((size_t)row << cb) + c0 + col;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (j < A.n_in) {
                const uint4 *src = reinterpret_cast<const uint4 *>(in + 8 * j);
                a = src[0];
                b = src[1];
                if (A.load_mode == 1) {
                    u32 m3 = (u32)(j % 3);
                    if (m3) {
                        fe v{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
                        v = fe_mulx<F>(v, from_param(m3 == 1 ? A.lk0 : A.lk1));
                        a = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
                        b = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
                    }
                }
            }
            u32 slot = (bitrev(row, r) << logT) + col;
            lo16[slot] = a;
            hi16[slot] = b;
        }
    } else {
        const size_t base = (hi_idx << (s0 + r)) + lo0;
        for (u32 e = tid; e < tile; e += nthr) {
            u32 col = e & (T - 1), mid = e >> logT;
            const uint4 *src = reinterpret_cast<const uint4 *>(in + 8 * (base + ((size_t)mid << s0) + col));
            lo16[e] = src[0];
            hi16[e] = src[1];
        }
    }
    

    // ---- R butterfly stages as radix-4 rounds (two stages per LDS round trip and per barrier; LDS writes
    //      are the slow direction on gfx950), plus one radix-2 round when R is odd.  One lane owns one radix-4
    //      group: elements mid00, mid00 | 2^u, mid00 | 2^(u+1), mid00 | 2^u | 2^(u+1).
    const size_t lo_x = FIRST ? 0 : (lo0 + (tid & (T - 1)));
    const u32 ngrp = tile >> 2;
    // twiddles of round u: stage t = s0 + u shares one (wA) between its two butterflies, stage t + 1 needs two (wB0 and,
    // n/4 further on, wB1).  They are fetched one round AHEAD: the L2 round trip of round u + 2's twiddles overlaps round
    // u's multiplications instead of following its barrier (every workgroup on the chip reaches that barrier at about
    // the same time, so nobody else has work to cover the latency).
    auto tw_addr = [&](int u, size_t &eA, size_t &eB0, size_t &eB1) {
        const int t = s0 + u;
        const u32 q = tid >> logT, low = q & ((1u << u) - 1);
        const size_t xm = ((size_t)low << s0) + lo_x;
        eA = xm << (L - t - 1);
        eB0 = xm << (L - t - 2);
        eB1 = eB0 + ((size_t)1 << (L - 2));
    };
    fe wA, wB0, wB1;
    if (R >= 2 && tid < ngrp) {
        size_t eA, eB0, eB1;
        tw_addr(0, eA, eB0, eB1);
        if (!FIRST) wA = fe_load(tw + 8 * eA);
        wB0 = fe_load(tw + 8 * eB0);
        wB1 = fe_load(tw + 8 * eB1);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u + 1 < R; u += 2) {
        fe nA, nB0, nB1;
        if (tid < ngrp) {
            if (u + 3 < R) {            // next radix-4 round exists: start its twiddle loads now
                size_t eA, eB0, eB1;
                tw_addr(u + 2, eA, eB0, eB1);
                nA = fe_load(tw + 8 * eA);
                nB0 = fe_load(tw + 8 * eB0);
                nB1 = fe_load(tw + 8 * eB1);
            }
            const u32 col = tid & (T - 1), q = tid >> logT;
            const u32 low = q & ((1u << u) - 1);
            const u32 mid00 = ((q >> u) << (u + 2)) | low;
            const u32 s00 = (mid00 << logT) + col, s01 = s00 + (T << u), s10 = s00 + (T << (u + 1)), s11 = s10 + (T << u);
            uint4 l0 = lo16[s00], h0 = hi16[s00], l1 = lo16[s01], h1 = hi16[s01];
            uint4 l2 = lo16[s10], h2 = hi16[s10], l3 = lo16[s11], h3 = hi16[s11];
            fe e0{{l0.x, l0.y, l0.z, l0.w, h0.x, h0.y, h0.z, h0.w}}, e1{{l1.x, l1.y, l1.z, l1.w, h1.x, h1.y, h1.z, h1.w}};
            fe e2{{l2.x, l2.y, l2.z, l2.w, h2.x, h2.y, h2.z, h2.w}}, e3{{l3.x, l3.y, l3.z, l3.w, h3.x, h3.y, h3.z, h3.w}};
            // lazy arithmetic (field.cuh): values stay in [0, 2p + d) inside and between the passes; the last pass's store
            // makes them canonical
            if (!(FIRST && u == 0)) {
                e1 = fe_mul_lazy<F>(e1, wA);
                e3 = fe_mul_lazy<F>(e3, wA);
            }
            fe a0 = fe_add_lazy<F>(e0, e1), a1 = fe_sub_lazy<F>(e0, e1), a2 = fe_add_lazy<F>(e2, e3), a3 = fe_sub_lazy<F>(e2, e3);
            a2 = fe_mul_lazy<F>(a2, wB0);
            a3 = fe_mul_lazy<F>(a3, wB1);
            e0 = fe_add_lazy<F>(a0, a2);
            e2 = fe_sub_lazy<F>(a0, a2);
            e1 = fe_add_lazy<F>(a1, a3);
            e3 = fe_sub_lazy<F>(a1, a3);
            lo16[s00] = make_uint4(e0.v[0], e0.v[1], e0.v[2], e0.v[3]);
            hi16[s00] = make_uint4(e0.v[4], e0.v[5], e0.v[6], e0.v[7]);
            lo16[s01] = make_uint4(e1.v[0], e1.v[1], e1.v[2], e1.v[3]);
            hi16[s01] = make_uint4(e1.v[4], e1.v[5], e1.v[6], e1.v[7]);
            lo16[s10] = make_uint4(e2.v[0], e2.v[1], e2.v[2], e2.v[3]);
            hi16[s10] = make_uint4(e2.v[4], e2.v[5], e2.v[6], e2.v[7]);
            lo16[s11] = make_uint4(e3.v[0], e3.v[1], e3.v[2], e3.v[3]);
            hi16[s11] = make_uint4(e3.v[4], e3.v[5], e3.v[6], e3.v[7]);
            wA = nA;
            wB0 = nB0;
            wB1 = nB1;
        }
        __syncthreads();
    }
    if (R & 1) {  // leftover radix-2 stage: tile/2 butterflies over tile/4 lanes
        constexpr int u = R - 1;
        const int t = s0 + u;
        const u32 nbf = tile >> 1;
        for (u32 bfl = tid; bfl < nbf; bfl += nthr) {
            const u32 col = bfl & (T - 1), q = bfl >> logT;
            const u32 low = q & ((1u << u) - 1);
            const u32 mid0 = ((q >> u) << (u + 1)) | low;
            const u32 s_a = (mid0 << logT) + col, s_b = s_a + (T << u);
            const size_t xm = ((size_t)low << s0) + (FIRST ? 0 : (lo0 + col));
            uint4 al = lo16[s_a], ah = hi16[s_a], bl = lo16[s_b], bh = hi16[s_b];
            fe a{{al.x, al.y, al.z, al.w, ah.x, ah.y, ah.z, ah.w}};
            fe b{{bl.x, bl.y, bl.z, bl.w, bh.x, bh.y, bh.z, bh.w}};
            if (!(FIRST && u == 0)) b = fe_mul_lazy<F>(b, fe_load(tw + 8 * (xm << (L - t - 1))));
            fe sm = fe_add_lazy<F>(a, b), d = fe_sub_lazy<F>(a, b);
            lo16[s_a] = make_uint4(sm.v[0], sm.v[1], sm.v[2], sm.v[3]);
            hi16[s_a] = make_uint4(sm.v[4], sm.v[5], sm.v[6], sm.v[7]);
            lo16[s_b] = make_uint4(d.v[0], d.v[1], d.v[2], d.v[3]);
            hi16[s_b] = make_uint4(d.v[4], d.v[5], d.v[6], d.v[7]);
        }
        __syncthreads();
    }

    // ---- store ----
    for (u32 e = tid; e < tile; e += nthr) {
        u32 col, mid;
        size_t x;
        if (FIRST) {
            mid = e & (rows - 1);
            col = e >> r;
            x = ((size_t)bitrev(c0 + col, L - r) << r) + mid;
        } else {
            col = e & (T - 1);
            mid = e >> logT;
            x = (hi_idx << (s0 + r)) + ((size_t)mid << s0) + lo0 + col;
        }
        u32 slot = (mid << logT) + col;
        uint4 a = lo16[slot], b = hi16[slot];
        if (A.store_mode) {
            fe v{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
            u32 m3 = A.store_mode == 2 ? (u32)(x % 3) : 0;
            v = fe_mulx<F>(v, from_param(m3 == 0 ? A.k0 : m3 == 1 ? A.k1 : A.k2));     // canonical product of a lazy value
            a = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            b = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        } else if (A.last) {
            fe v{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
            v = fe_reduce_lazy<F>(v);
            a = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            b = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + 8 * x);
        dst[0] = a;
        dst[1] = b;
    }
}

// ---- the same pass on the carry-free 9 x 29-bit field layer (field9.cuh) -----------------------------------------------
// A butterfly is one multiplication and two additions: on the 8 x 32 layer 248 + 2 x ~35 instructions, here 165 + 2 x 9 plus
// one "fold" per element and round.  LDS holds nine signed limbs per element (three planes: limbs 0-3, 4-7, 8): nothing is
// repacked between the stages of a pass.  Twiddles come from a second table flavour, omega^e in M9 form as raw limbs in the same
// three planes, so data keeps the caller's form: (a 2^256)(w 2^261) / 2^261 = a w 2^256.
// Bounds: only the LIMBS have to stay small between the stages -- a twiddle is below p < 2^255, so a multiplication tolerates
// |value| < 2^261 on the data side.  A radix-4 round adds two products (each below 2^256.3 in magnitude) to an element, so
// after the five rounds of a 10-stage pass |value| < 2^256.4 + 5 x 2^257.3 < 2^260; limb 8 absorbs the growth.  The value is
// folded once, when the element leaves the pass: q = round(value / 2^254) from the top limb, minus q p read from a 129-entry
// LDS table -> |value| < 2^253.1.
// Carry passes (round 4): the CONSUMER normalises, and only what it adds without multiplying.  A round's outputs are
//     o = (e0 +- e1 wA) +- (e2 +- e3 wA) wB      with every product's limbs in [0, 2^29]   (field9.cuh)
// so with e0 normalised (limbs 0..7 in [0, 2^29)) every output limb lies in (-2^30, 3 x 2^29) -- call that RAW.  Next round:
//   * e1, e3 are multiplied as they are: a column of the multiplier holds at most 9 x (3 x 2^29)(2^29) = 27 x 2^58 of products
//     plus < 3.01 x 2^58 of reduction terms plus a carry below 2^34 -- under 2^63;
//   * e2 meets a product before ITS multiplication (e2 +- e3 wA): raw it could reach 4 x 2^29 (36 x 2^58 per column: too much),
//     so it takes a carry pass first -- then |e2 +- e3 wA| < 2^30;
//   * e0 is never multiplied: it takes a carry pass so that the outputs are RAW again.
// Two carry passes per radix-4 group and round instead of four (one per output), none in a pass's first round (its inputs come
// unpacked from memory) and none after its last (the fold / the store factor's multiplication take RAW limbs): 8 instead of 20
// per lane in a 10-stage pass, ~290 of ~5600 instructions.
struct Tw9 {
    const uint4 *a, *b;
    const u32 *c;
};
__device__ __forceinline__ fe9 tw9_load(const Tw9 &t, size_t e) {
    const uint4 x = t.a[e], y = t.b[e];
    return fe9{{(i32)x.x, (i32)x.y, (i32)x.z, (i32)x.w, (i32)y.x, (i32)y.y, (i32)y.z, (i32)y.w, (i32)t.c[e]}};
}
// entry `base + xm` with `base` wave-uniform (a stage's first entry) and xm < 2^27 per lane: the lane offset stays a 32-bit byte
// offset beside a scalar base (global_load ... v_off, s[base:base+1]) instead of three 64-bit address computations per twiddle
__device__ __forceinline__ fe9 tw9_load32(const Tw9 &t, size_t base, u32 xm) {
    const char *pa = reinterpret_cast<const char *>(t.a + base), *pb = reinterpret_cast<const char *>(t.b + base),
               *pc = reinterpret_cast<const char *>(t.c + base);
    const u32 o16 = xm << 4, o4 = xm << 2;
    const uint4 x = *reinterpret_cast<const uint4 *>(pa + o16), y = *reinterpret_cast<const uint4 *>(pb + o16);
    const u32 z = *reinterpret_cast<const u32 *>(pc + o4);
    return fe9{{(i32)x.x, (i32)x.y, (i32)x.z, (i32)x.w, (i32)y.x, (i32)y.y, (i32)y.z, (i32)y.w, (i32)z}};
}
struct Lds9 {
    uint4 *a, *b;
    u32 *c;
    const i32 *qp;     // q p for q = -64 .. 64, 12 words apart
};
__device__ __forceinline__ fe9 lds9_get(const Lds9 &l, u32 s) {
    const uint4 x = l.a[s], y = l.b[s];
    return fe9{{(i32)x.x, (i32)x.y, (i32)x.z, (i32)x.w, (i32)y.x, (i32)y.y, (i32)y.z, (i32)y.w, (i32)l.c[s]}};
}
__device__ __forceinline__ void lds9_put(const Lds9 &l, u32 s, const fe9 &v) {
    l.a[s] = make_uint4((u32)v.v[0], (u32)v.v[1], (u32)v.v[2], (u32)v.v[3]);
    l.b[s] = make_uint4((u32)v.v[4], (u32)v.v[5], (u32)v.v[6], (u32)v.v[7]);
    l.c[s] = (u32)v.v[8];
}
__device__ __forceinline__ fe9 ntt_fold9(const fe9 &v, const i32 *qp) {
    const i32 q = (v.v[8] + (1 << 21)) >> 22;            // |value| < 2^260: q in [-64, 64]
    const uint4 *e = reinterpret_cast<const uint4 *>(qp + 12 * (q + 64));
    const uint4 x = e[0], y = e[1];
    const i32 z = qp[12 * (q + 64) + 8];
    fe9 r;
    r.v[0] = v.v[0] - (i32)x.x; r.v[1] = v.v[1] - (i32)x.y; r.v[2] = v.v[2] - (i32)x.z; r.v[3] = v.v[3] - (i32)x.w;
    r.v[4] = v.v[4] - (i32)y.x; r.v[5] = v.v[5] - (i32)y.y; r.v[6] = v.v[6] - (i32)y.z; r.v[7] = v.v[7] - (i32)y.w;
    r.v[8] = v.v[8] - z;
    return fe9_norm(r);
}
// canonical packed value of a FOLDED element: |value| < 2^253 + 2^132 (ntt_fold9) and v is NORMALISED (the fold ends in a carry
// pass), so fe9_pack's shifts and ors make the 256-bit TWO'S COMPLEMENT word of the value and its sign is bit 255: add p exactly
// when that bit is set, as one 8-word carry chain on the packed words (p = 2^254 + t has five non-zero words: ~14 instructions).
// Round 3 added p, carried, subtracted p again, carried again and selected (135 instructions per element); round 4 selected on
// limb 8's sign and spent one more 24-instruction carry pass on the limbs before packing (~70); this is ~45.
template <int F> __device__ __forceinline__ fe ntt_canonical_folded9(const fe9 &v) {
    const fe w = fe9_pack(v);
    const u32 neg = (u32)((i32)w.v[7] >> 31);     // all ones iff value < 0
    fe r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 co;
        r.v[i] = __builtin_addc(w.v[i], mod_limb<F>(i) & neg, c, &co);
        c = co;
    }
    return r;                                     // value in [0, p)
}
// What leaves a pass that is not the last: the folded value as a 256-bit TWO'S COMPLEMENT word (|value| < 2^253.1 fits with room
// to spare; fe9_pack's shifts and ors are already that for a negative limb 8), read back by ntt_unpack_signed9 with an arithmetic
// shift for limb 8.  The intermediate vector never leaves the transform, and adding p + a second carry pass to make it
// non-negative is ~30 instructions per element saved.
__device__ __forceinline__ fe ntt_pack_signed9(const fe9 &v) { return fe9_pack(v); }            // v normalised
__device__ __forceinline__ fe9 ntt_unpack_signed9(const fe &a) {
    fe9 r = fe9_unpack(a);
    r.v[8] = (i32)a.v[7] >> 8;                    // bits 232..255, sign-extended
    return r;
}

template <int F, int R, bool FIRST>
__global__ void __launch_bounds__(1024) ntt_pass9(const u32 *__restrict__ in, u32 *__restrict__ out, Tw9 tw, PassArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int r = R;
    const int logT = A.logT, L = A.L, s0 = A.s0;
    const u32 T = 1u << logT, rows = 1u << r, tile = rows << logT;
    Lds9 S;
    S.a = lds;
    S.b = lds + tile;
    S.c = reinterpret_cast<u32 *>(lds + 2 * tile);
    i32 *qp_w = reinterpret_cast<i32 *>(S.c + tile);
    S.qp = qp_w;
    const u32 tid = threadIdx.x, nthr = blockDim.x;
    for (u32 j = tid; j < 129; j += nthr) {     // q p as signed limbs (|q| <= 64: every limb of the product fits 36 bits before the carry pass)
        const i64 q = (i64)j - 64;
        const fe9 pk = fe9_p_shl<F>(0);
        i64 c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const i64 t = q * pk.v[i] + c;
            qp_w[12 * j + i] = (i32)(t & (i64)M29);
            c = t >> 29;
        }
        qp_w[12 * j + 8] = (i32)(q * pk.v[8] + c);
    }
    size_t hi_idx = 0, lo0 = 0;
    u32 c0 = 0;
    if (FIRST) {
        c0 = blockIdx.x << logT;
    } else {
        const u32 tile_id = ntt_tile_of_block(blockIdx.x, gridDim.x, s0, logT);
        u32 tiles_per_hi = 1u << (s0 - logT);
        hi_idx = tile_id / tiles_per_hi;
        lo0 = (size_t)(tile_id % tiles_per_hi) << logT;
    }
    // The first round takes its four elements straight from global memory and the last one stores straight to it: no load-all /
    // barrier / store-all phases, a wave starts multiplying as soon as ITS loads are back, and two of the LDS round trips
    // disappear.  An ODD stage count (round 5: 11- and 12-stage passes make 2^21 .. 2^24 two-pass transforms) opens with a
    // radix-2 round on the same four elements per lane -- the first HALF of a radix-4 round: both pairs (rows 4q, 4q + 1 and
    // 4q + 2, 4q + 3) share the stage's twiddle -- and continues with radix-4 rounds from stage 1; the transform's very first
    // stage multiplies by omega^0 only, so there that round is four loads, four additions and four LDS writes.
    constexpr bool FUSE_LOAD = R >= 2, FUSE_STORE = R >= 2;
    constexpr int U0 = (R >= 3 && (R & 1)) ? 1 : 0;      // first stage of the radix-4 rounds
    const u32 ngrp = tile >> 2;
    // the element that sits in LDS row `row` (after the first pass's bit reversal), column `col`
    auto load_elem = [&](u32 row, u32 col, const fe9 &lk0, const fe9 &lk1) -> fe9 {
        if (FIRST) {
            const size_t j = ((size_t)bitrev(row, r) << (L - r)) + c0 + col;
            fe9 v = fe9_zero();
            if (j < A.n_in) {
                v = fe9_unpack(fe_load(in + 8 * j));
                if (A.load_mode == 1) {
                    const u32 m3 = (u32)(j % 3);
                    if (m3) v = fe9_mul<F>(v, m3 == 1 ? lk0 : lk1);
                }
            }
            return v;
        }
        return ntt_unpack_signed9(fe_load(in + 8 * ((hi_idx << (s0 + r)) + lo0 + ((size_t)row << s0) + col)));      // ntt_pack_signed9 wrote it
    };
    auto load_factors = [&](fe9 &lk0, fe9 &lk1) {
        lk0 = lk1 = fe9_zero();
        if (FIRST && A.load_mode == 1) {
            lk0 = fe9_from_r256<F>(from_param(A.lk0));
            lk1 = fe9_from_r256<F>(from_param(A.lk1));
        }
    };
    auto store_factors = [&](fe9 &k0, fe9 &k1, fe9 &k2) {
        k0 = k1 = k2 = fe9_zero();
        if (A.store_mode) {
            k0 = fe9_from_r256<F>(from_param(A.k0));
            if (A.store_mode == 2) {
                k1 = fe9_from_r256<F>(from_param(A.k1));
                k2 = fe9_from_r256<F>(from_param(A.k2));
            }
        }
    };
    // the element of LDS row `mid`, column `col` leaves the pass
    auto store_elem = [&](u32 mid, u32 col, fe9 v, const fe9 &k0, const fe9 &k1, const fe9 &k2) {
        const size_t x = FIRST ? ((size_t)bitrev(c0 + col, L - r) << r) + mid : (hi_idx << (s0 + r)) + ((size_t)mid << s0) + lo0 + col;
        if (!A.store_mode) v = ntt_fold9(v, S.qp);          // (a multiplication by the store factor takes the unfolded value)
        fe w;
        if (A.store_mode) {
            const u32 m3 = A.store_mode == 2 ? (u32)(x % 3) : 0;
            w = fe9_canonical_small<F>(fe9_mul<F>(v, m3 == 0 ? k0 : m3 == 1 ? k1 : k2));
        } else if (A.last) {
            w = ntt_canonical_folded9<F>(v);
        } else {
            w = ntt_pack_signed9(v);
        }
        fe_store(out + 8 * x, w);
    };
    if (!FUSE_LOAD) {
        fe9 lk0, lk1;
        load_factors(lk0, lk1);
        for (u32 e = tid; e < tile; e += nthr) lds9_put(S, e, load_elem(e >> logT, e & (T - 1), lk0, lk1));
    }
    // group q of round u in column col: rows mid00 + {0, 2^u, 2^(u+1), 3 2^u}; lanes take col fastest, except in a transposing
    // (first-pass) fused store, where q runs fastest so that a wave writes one contiguous run of its column
    auto lane_of = [&](int u, u32 &q, u32 &col) {
        if (FIRST && FUSE_STORE && u == R - 2) {
            q = tid & ((1u << (r - 2)) - 1);
            col = tid >> (r - 2);
        } else {
            q = tid >> logT;
            col = tid & (T - 1);
        }
    };
    // stage-major table: the 2^t twiddles of stage t, omega^(xm 2^(L-t-1)) for xm < 2^t, sit contiguously at 2^t - 1 + xm, so
    // the T lanes of a tile row read T consecutive entries (one 128-byte run per plane) instead of entries 2^(L-t-1) apart
    // the three twiddles of round u: which = 0 (wA, entry 2^t - 1 + xm), 1 (wB0, 2^(t+1) - 1 + xm), 2 (wB1, 2^t further on).  The host
    // routes transforms beyond 2^28 to the 8 x 32 kernel, so every in-stage index xm is below 2^27 here and stays a 32-bit lane
    // offset beside the stage's scalar base (tw9_load32).
    auto tw_get = [&](int u, int which) -> fe9 {
        u32 q, col;
        lane_of(u, q, col);
        const int t = s0 + u;
        const u32 low = q & ((1u << u) - 1);
        const u32 xm = (low << s0) + (FIRST ? 0u : (u32)lo0 + col);
        const size_t base = which == 0 ? (((size_t)1 << t) - 1) : which == 1 ? (((size_t)2 << t) - 1) : (((size_t)3 << t) - 1);
        return tw9_load32(tw, base, xm);
    };
    // the first twiddle of a round (needed at once) is fetched one round ahead; the other two are requested at the top of the
    // round and first used two multiplications later
    fe9 wA = fe9_zero();
    if (R >= 2 && tid < ngrp && !FIRST) wA = tw_get(0, 0);
    __syncthreads();                                   // the q p table (and, unfused, the tile) is in LDS
    if (U0) {                                          // odd stage count: stage 0 as a radix-2 round on four elements per lane
        if (tid < ngrp) {
            const fe9 nA = tw_get(1, 0);
            u32 q, col;
            lane_of(0, q, col);
            const u32 mid00 = q << 2, s00 = (mid00 << logT) + col;
            fe9 lk0, lk1;
            load_factors(lk0, lk1);
            const fe9 e0 = load_elem(mid00, col, lk0, lk1), e2 = load_elem(mid00 + 2, col, lk0, lk1);
            fe9 e1 = load_elem(mid00 + 1, col, lk0, lk1), e3 = load_elem(mid00 + 3, col, lk0, lk1);
            if (!FIRST) {
                e1 = fe9_mul<F>(e1, wA);
                e3 = fe9_mul<F>(e3, wA);
            }
            // unpacked element (limbs in [0, 2^29), limb 8 small) +- product or unpacked element: RAW
            lds9_put(S, s00, fe9_add(e0, e1));
            lds9_put(S, s00 + T, fe9_sub(e0, e1));
            lds9_put(S, s00 + 2 * T, fe9_add(e2, e3));
            lds9_put(S, s00 + 3 * T, fe9_sub(e2, e3));
            wA = nA;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = U0; u + 1 < R; u += 2) {
        if (tid < ngrp) {
            const fe9 wB0 = (FIRST && u == 0) ? fe9_zero() : tw_get(u, 1), wB1 = tw_get(u, 2);
            fe9 nA = fe9_zero();
            if (u + 3 < R) nA = tw_get(u + 2, 0);
            u32 q, col;
            lane_of(u, q, col);
            const u32 low = q & ((1u << u) - 1);
            const u32 mid00 = ((q >> u) << (u + 2)) | low;
            const u32 s00 = (mid00 << logT) + col, s01 = s00 + (T << u), s10 = s00 + (T << (u + 1)), s11 = s10 + (T << u);
            fe9 e0, e1, e2, e3;
            if (FUSE_LOAD && u == 0) {                 // (even stage counts: an odd one has its elements in LDS by now)
                fe9 lk0, lk1;
                load_factors(lk0, lk1);
                e0 = load_elem(mid00, col, lk0, lk1);
                e1 = load_elem(mid00 + 1, col, lk0, lk1);
                e2 = load_elem(mid00 + 2, col, lk0, lk1);
                e3 = load_elem(mid00 + 3, col, lk0, lk1);
            } else {
                // RAW limbs in LDS (header comment): the two elements that are added before anything multiplies them take the carry pass
                e0 = fe9_norm(lds9_get(S, s00)), e1 = lds9_get(S, s01), e2 = fe9_norm(lds9_get(S, s10)), e3 = lds9_get(S, s11);
            }
            if (!(FIRST && u == 0)) {
                e1 = fe9_mul<F>(e1, wA);
                e3 = fe9_mul<F>(e3, wA);
            }
            const fe9 a0 = fe9_add(e0, e1), a1 = fe9_sub(e0, e1);
            // the transform's first round: stage 1's twiddle for the pair (e2 + e3) is omega^0 for every group (stage-major entry 1) --
            // a carry pass stands in for that multiplication by one (limbs back in [0, 2^29): the outputs stay RAW)
            const fe9 a2 = (FIRST && u == 0) ? fe9_norm(fe9_add(e2, e3)) : fe9_mul<F>(fe9_add(e2, e3), wB0);
            const fe9 a3 = fe9_mul<F>(fe9_sub(e2, e3), wB1);
            const fe9 o00 = fe9_add(a0, a2), o10 = fe9_sub(a0, a2);                   // RAW: limbs in (-2^30, 3 x 2^29)
            const fe9 o01 = fe9_add(a1, a3), o11 = fe9_sub(a1, a3);
            if (FUSE_STORE && u == R - 2) {
                fe9 k0, k1, k2;
                store_factors(k0, k1, k2);
                store_elem(mid00, col, o00, k0, k1, k2);
                store_elem(mid00 + (1u << u), col, o01, k0, k1, k2);
                store_elem(mid00 + (2u << u), col, o10, k0, k1, k2);
                store_elem(mid00 + (3u << u), col, o11, k0, k1, k2);
            } else {
                lds9_put(S, s00, o00);
                lds9_put(S, s10, o10);
                lds9_put(S, s01, o01);
                lds9_put(S, s11, o11);
            }
            wA = nA;
        }
        if (!(FUSE_STORE && u == R - 2)) __syncthreads();
    }
    if (R == 1) {                                      // a lone stage (the tail of a plan whose stage count does not split evenly)
        constexpr int u = R - 1;
        const int t = s0 + u;
        const u32 nbf = tile >> 1;
        for (u32 bfl = tid; bfl < nbf; bfl += nthr) {
            const u32 col = bfl & (T - 1), q = bfl >> logT;
            const u32 low = q & ((1u << u) - 1);
            const u32 mid0 = ((q >> u) << (u + 1)) | low;
            const u32 s_a = (mid0 << logT) + col, s_b = s_a + (T << u);
            const size_t xm = ((size_t)low << s0) + (FIRST ? 0 : (lo0 + col));
            fe9 a = lds9_get(S, s_a), b = lds9_get(S, s_b);
            if (!(FIRST && u == 0)) b = fe9_mul<F>(b, tw9_load32(tw, ((size_t)1 << t) - 1, (u32)xm));
            lds9_put(S, s_a, fe9_norm(fe9_add(a, b)));
            lds9_put(S, s_b, fe9_norm(fe9_sub(a, b)));
        }
        __syncthreads();
    }
    // ---- store (passes that end in a radix-2 round) ----
    if (!FUSE_STORE) {
        fe9 k0, k1, k2;
        store_factors(k0, k1, k2);
        for (u32 e = tid; e < tile; e += nthr) {
            u32 col, mid;
            if (FIRST) {
                mid = e & (rows - 1);
                col = e >> r;
            } else {
                col = e & (T - 1);
                mid = e >> logT;
            }
            store_elem(mid, col, lds9_get(S, (mid << logT) + col), k0, k1, k2);
        }
    }
}

// twiddle table, M9 flavour: omega^e in M9 form as raw limbs, three planes (limbs 0-3 | 4-7 | 8), STAGE-MAJOR: stage t's
// entries omega^(xm 2^(L-t-1)), xm < 2^t, at index 2^t - 1 + xm (2^L - 1 entries in all; omega^e is stored once for every
// stage whose stride divides e).
template <int F>
__global__ void __launch_bounds__(256) ntt_twiddles9(uint4 *__restrict__ pa, uint4 *__restrict__ pb, u32 *__restrict__ pc, feparam omega_p,
                                                     feparam step_p, u32 T, size_t count, int L) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    fe omega = from_param(omega_p), step = from_param(step_p);
    fe cur = fe_one<F>();
    for (int b = 31 - __clz(t | 1); b >= 0; --b) {
        cur = fe_sqr<F>(cur);
        if ((t >> b) & 1) cur = fe_mulx<F>(cur, omega);
    }
    for (size_t e = t; e < count; e += T) {
        const fe9 v = fe9_unpack(fe_mulx<F>(cur, fe_k32<F>()));
        const uint4 va = make_uint4((u32)v.v[0], (u32)v.v[1], (u32)v.v[2], (u32)v.v[3]);
        const uint4 vb = make_uint4((u32)v.v[4], (u32)v.v[5], (u32)v.v[6], (u32)v.v[7]);
        // stages st = L - 1 down to the first whose stride 2^(L-st-1) no longer divides e (e = 0: every stage)
        for (int st = L - 1; st >= 0; --st) {
            const int sh = L - st - 1;
            if (e & ((((size_t)1) << sh) - 1)) break;
            const size_t idx = (((size_t)1 << st) - 1) + (e >> sh);
            pa[idx] = va;
            pb[idx] = vb;
            pc[idx] = (u32)v.v[8];
        }
        cur = fe_mulx<F>(cur, step);
    }
}

// elementwise multiply for the degenerate log_n = 0 case
template <int F> __global__ void ntt_scale1(u32 *a, feparam k) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fe_store(a, fe_mulx<F>(fe_load(a), from_param(k)));
}

// ---- host: twiddle cache + pass plan ---------------------------------------------------------------
struct TwKey {
    int dev, field, L;
    int flavour;     // 0: omega^e in the reference's Montgomery form, 32 B each (also read by the curve-point FFT); 1: M9 raw limbs, 36 B
    u64 w[4];
    bool operator==(const TwKey &o) const {
        return dev == o.dev && field == o.field && L == o.L && flavour == o.flavour && memcmp(w, o.w, 32) == 0;
    }
    bool operator<(const TwKey &o) const {
        if (dev != o.dev) return dev < o.dev;
        if (field != o.field) return field < o.field;
        if (L != o.L) return L < o.L;
        if (flavour != o.flavour) return flavour < o.flavour;
        return memcmp(w, o.w, 32) < 0;
    }
};
struct TwEntry {
    void *d = nullptr;
    hipEvent_t ready = nullptr;
    ~TwEntry() {
        if (d) (void)hipFree(d);
        if (ready) (void)hipEventDestroy(ready);
    }
};
struct NttContext {
    std::mutex mu;
    std::map<TwKey, std::shared_ptr<TwEntry>> cache;
    std::list<TwKey> lru;
    size_t cache_bytes = 0;
    std::map<std::pair<int, hipStream_t>, DevBuf> tmp, stage;
    bool attr_set = false;
};
static NttContext &ntt_ctx() {
    static NttContext c;
    return c;
}
static const size_t kTwCacheBytes = (size_t)6 << 30;  // HBM is 288 GB: keep tables around
void ntt_release_workspaces() {   // h2_trim: scratch vectors and cached twiddle tables (rebuilt on the next transform)
    NttContext &cx = ntt_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    for (auto &kv : cx.tmp) kv.second.release();
    for (auto &kv : cx.stage) kv.second.release();
    cx.cache.clear();
    cx.lru.clear();
    cx.cache_bytes = 0;
}

// returns the device table omega^0..omega^(n/2-1); builds it on `st` when missing
static int get_twiddles(NttContext &cx, int field, int L, const u64 omega_m[4], hipStream_t st, std::shared_ptr<TwEntry> &out,
                        int flavour = 0) {
    TwKey key;
    (void)hipGetDevice(&key.dev);
    key.field = field;
    key.L = L;
    key.flavour = flavour;
    const size_t esz = flavour ? 72 : 32;     // M9 flavour: stage-major, 2^L - 1 entries of 36 B = 72 B per omega^e, e < n / 2
    memcpy(key.w, omega_m, 32);
    auto it = cx.cache.find(key);
    if (it != cx.cache.end()) {
        cx.lru.remove(key);
        cx.lru.push_front(key);
        out = it->second;
        // order this stream behind the build
        H2_HIP(hipStreamWaitEvent(st, out->ready, 0));
        return H2_OK;
    }
    size_t count = L >= 1 ? ((size_t)1 << (L - 1)) : 1;
    auto ent = std::make_shared<TwEntry>();
    H2_HIP(hipMalloc(&ent->d, count * esz));
    H2_HIP(hipEventCreateWithFlags(&ent->ready, hipEventDisableTiming));
    u32 T = (u32)std::min<size_t>(count, 1u << 16);
    u64 step[4];
    memcpy(step, omega_m, 32);
    for (u32 s = 1; s < T; s <<= 1) host_mul(field, step, step, step);  // omega^T, T a power of two
    dim3 grid((T + 255) / 256), block(256);
    if (flavour) {
        uint4 *pa = (uint4 *)ent->d, *pb = pa + 2 * count;
        u32 *pc = (u32 *)(pb + 2 * count);
        if (field == H2_FP) hipLaunchKernelGGL((ntt_twiddles9<FP>), grid, block, 0, st, pa, pb, pc, to_param(omega_m), to_param(step), T, count, L);
        else hipLaunchKernelGGL((ntt_twiddles9<FQ>), grid, block, 0, st, pa, pb, pc, to_param(omega_m), to_param(step), T, count, L);
    } else if (field == H2_FP)
        hipLaunchKernelGGL((ntt_twiddles<FP>), grid, block, 0, st, (u32 *)ent->d, to_param(omega_m), to_param(step), T, count);
    else
        hipLaunchKernelGGL((ntt_twiddles<FQ>), grid, block, 0, st, (u32 *)ent->d, to_param(omega_m), to_param(step), T, count);
    H2_HIP(hipGetLastError());
    H2_HIP(hipEventRecord(ent->ready, st));
    cx.cache[key] = ent;
    cx.lru.push_front(key);
    cx.cache_bytes += count * esz;
    while (cx.cache_bytes > kTwCacheBytes && cx.lru.size() > 1) {
        TwKey old = cx.lru.back();
        cx.lru.pop_back();
        auto o = cx.cache.find(old);
        if (o != cx.cache.end()) {
            // entries still referenced by in-flight work stay alive through the shared_ptr held by the caller
            H2_HIP(hipEventSynchronize(o->second->ready));
            H2_HIP(hipDeviceSynchronize());
            cx.cache_bytes -= (old.L >= 1 ? ((size_t)1 << (old.L - 1)) : 1) * (old.flavour ? 72 : 32);
            cx.cache.erase(o);
        }
    }
    out = ent;
    return H2_OK;
}

static bool ntt_on_fe9() {
    static const bool on = [] { const char *e = ab_env("H2_NTT_FE9"); return !(e && atoi(e) == 0); }();
    return on;
}
// the carry-free passes keep in-stage twiddle indices in 32-bit lane offsets (tw9_load32): transforms up to 2^28; beyond that
// (16 GiB vectors and up) the 8 x 32 kernel with its 32-byte table entries takes over
static bool ntt_use_fe9(int L) { return ntt_on_fe9() && L <= 28; }
template <int F, int R, bool FIRST>
static int launch_pass_t(const PassArgs &A, unsigned tiles, u32 threads, size_t lds, hipStream_t st, const u32 *src, u32 *dst,
                         const u32 *tw) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (ntt_use_fe9(A.L)) {
        static bool attr9[16] = {false};     // per DEVICE (the attribute is): a process driving several GPUs sets it on each; callers hold cx.mu
        if (!attr9[dev & 15]) {
            H2_HIP(hipFuncSetAttribute((const void *)ntt_pass9<F, R, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr9[dev & 15] = true;
        }
        const size_t count = A.L >= 1 ? ((size_t)1 << (A.L - 1)) : 1;
        Tw9 t9;
        t9.a = (const uint4 *)tw;
        t9.b = t9.a + 2 * count;
        t9.c = (const u32 *)(t9.b + 2 * count);
        const size_t lds9 = lds / 32 * 36 + 129 * 48;
        hipLaunchKernelGGL((ntt_pass9<F, R, FIRST>), dim3(tiles), dim3(threads), lds9, st, src, dst, t9, A);
        return H2_OK;
    }
    static bool attr[16] = {false};  // raise the dynamic-LDS cap once per instantiation and device
    if (!attr[dev & 15]) {
        H2_HIP(hipFuncSetAttribute((const void *)ntt_pass<F, R, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr[dev & 15] = true;
    }
    hipLaunchKernelGGL((ntt_pass<F, R, FIRST>), dim3(tiles), dim3(threads), lds, st, src, dst, tw, A);
    return H2_OK;
}
template <int F, bool FIRST>
static int launch_pass_r(const PassArgs &A, unsigned tiles, u32 threads, size_t lds, hipStream_t st, const u32 *src, u32 *dst,
                         const u32 *tw) {
    switch (A.r) {
        case 1: return launch_pass_t<F, 1, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 2: return launch_pass_t<F, 2, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 3: return launch_pass_t<F, 3, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 4: return launch_pass_t<F, 4, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 5: return launch_pass_t<F, 5, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 6: return launch_pass_t<F, 6, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 7: return launch_pass_t<F, 7, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 8: return launch_pass_t<F, 8, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 9: return launch_pass_t<F, 9, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 10: return launch_pass_t<F, 10, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 11: return launch_pass_t<F, 11, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
        case 12: return launch_pass_t<F, 12, FIRST>(A, tiles, threads, lds, st, src, dst, tw);
    }
    return H2_ERR_ARGS;
}
static int launch_pass(int field, const PassArgs &A, unsigned tiles, u32 threads, size_t lds, hipStream_t st, const u32 *src,
                       u32 *dst, const u32 *tw) {
    if (field == H2_FP)
        return A.first ? launch_pass_r<FP, true>(A, tiles, threads, lds, st, src, dst, tw)
                       : launch_pass_r<FP, false>(A, tiles, threads, lds, st, src, dst, tw);
    return A.first ? launch_pass_r<FQ, true>(A, tiles, threads, lds, st, src, dst, tw)
                   : launch_pass_r<FQ, false>(A, tiles, threads, lds, st, src, dst, tw);
}

int ntt_twiddle_table(int field, int L, const uint64_t omega_mont[4], hipStream_t st, const uint32_t **d_tw) {
    NttContext &cx = ntt_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    std::shared_ptr<TwEntry> tw;
    int rc = get_twiddles(cx, field, L, omega_mont, st, tw);
    if (rc != H2_OK) return rc;
    *d_tw = (const uint32_t *)tw->d;   // stays alive in the cache (evicted only beyond 6 GiB of tables)
    return H2_OK;
}

struct NttJob {
    int field;
    unsigned L;
    const void *d_in;   // first-pass source (n_in valid elements)
    void *d_out;        // final destination (2^L elements)
    size_t n_in;
    int load_mode, store_mode;
    u64 omega[4];       // Montgomery
    u64 lk0[4], lk1[4];           // load multipliers (Montgomery)
    u64 sk0[4], sk1[4], sk2[4];   // store multipliers (Montgomery)
    int plan = 0;                 // 0: fewest passes (one transform owns the chip); 1: small tiles, for concurrent transforms
};

static int ntt_run(const NttJob &J, hipStream_t st) {
    NttContext &cx = ntt_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    if (!cx.attr_set) {
        cx.attr_set = true;
    }
    const int L = (int)J.L;
    if (L == 0) {
        if (J.d_in != J.d_out) H2_HIP(hipMemcpyAsync(J.d_out, J.d_in, 32, hipMemcpyDeviceToDevice, st));
        if (J.store_mode) {
            if (J.field == H2_FP) hipLaunchKernelGGL((ntt_scale1<FP>), dim3(1), dim3(64), 0, st, (u32 *)J.d_out, to_param(J.sk0));
            else hipLaunchKernelGGL((ntt_scale1<FQ>), dim3(1), dim3(64), 0, st, (u32 *)J.d_out, to_param(J.sk0));
        }
        return H2_OK;
    }
    std::shared_ptr<TwEntry> tw;
    int rc = get_twiddles(cx, J.field, L, J.omega, st, tw, ntt_use_fe9(L) ? 1 : 0);
    if (rc != H2_OK) return rc;

    // pass plan: ceil(L / maxr) passes, stages spread evenly (H2_NTT_MAXR / H2_NTT_LOGT / H2_NTT_LDS: tuning sweeps only).
    // Up to 10 stages per pass with 128 KiB tiles: 2^20 runs as TWO passes of 10 stages (one workgroup per CU, the whole
    // vector resident in LDS across the chip) instead of three of 7, 7, 6 -- 0.137 -> 0.128 ms; 2^22 still needs three.
    // Plan 0 (a transform alone): up to 10 stages per pass with 128 KiB tiles -- 2^20 runs as TWO passes of 10 stages (one
    // workgroup per CU, the whole vector resident in LDS across the chip) instead of three of 7, 7, 6: 0.137 -> 0.128 ms.
    // Plan 1 (the batch entry points: independent column transforms on internal streams): at most 8 stages and 64 KiB, so
    // workgroups of several transforms share a CU and one column's load / store phases hide under another's butterflies
    // (0.105 ms per 2^20 transform over 3 streams, against 0.131 with plan 0).  H2_NTT_MAXR / _LOGT / _LDS: sweeps only.
    // Round 5: 11- and 12-stage passes exist (2048 rows x 2 columns / 4096 rows x 1 column of nine-limb elements = 147 KiB, 1024
    // lanes; odd stage counts open with a radix-2 round), so 2^21 .. 2^24 CAN run as two passes -- and measured on the same box
    // (profiles/r05_ntt_two_pass_ab.txt) that is SLOWER: 2^22 as 11 + 11 0.398 ms against 0.342 as 8 + 8 + 6, 2^24 as 12 + 12 2.09
    // against 1.45.  The passes are issue-bound, not byte-bound: two passes carry ~10 800 instructions per lane-quadruple against
    // ~11 170 for three (3 % fewer), while their 64- / 32-byte rows and one-workgroup-per-CU tiles lose more than that to the
    // memory phases no second workgroup covers.  The default stays at 10 stages; H2_NTT_MAXR=11 / 12 reproduces the A/B.
    static const int env_maxr = [] { const char *e = ab_env("H2_NTT_MAXR"); int v = e ? atoi(e) : 0; return v >= 1 && v <= 12 ? v : 0; }();
    static const int want_logT = [] { const char *e = ab_env("H2_NTT_LOGT"); int v = e ? atoi(e) : 3; return v >= 0 && v <= 5 ? v : 3; }();
    static const u32 env_lds = [] { const char *e = ab_env("H2_NTT_LDS"); int v = e ? atoi(e) : 131072; return (u32)(v >= 32768 && v <= 131072 ? v : 131072); }();
    const int dflt_maxr = env_maxr ? env_maxr : 10;
    const int maxr = J.plan == 1 ? std::min(dflt_maxr, 8) : dflt_maxr;
    const u32 lds_cap = J.plan == 1 ? std::min<u32>(env_lds, 65536u) : env_lds;
    const int P = (L + maxr - 1) / maxr;
    int stages[40];
    for (int i = 0; i < P; ++i) stages[i] = L / P + (i < L % P ? 1 : 0);
    // passes with an even stage count fuse their loads and stores into the first and last radix-4 round: pair up odd counts
    for (int i = 0; i < P; ++i) {
        if (!(stages[i] & 1)) continue;
        for (int j = i + 1; j < P; ++j)
            if ((stages[j] & 1) && stages[i] + 1 <= maxr && stages[j] >= 2) {
                stages[i] += 1;
                stages[j] -= 1;
                break;
            }
    }
    const size_t n = (size_t)1 << L;
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *tmp = nullptr;
    if (P > 1 && J.d_in == J.d_out) {
        DevBuf &tb = cx.tmp[std::make_pair(dev, st)];
        if ((rc = tb.reserve(n * 32)) != H2_OK) return rc;
        tmp = tb.ptr;
    }
    int s0 = 0;
    for (int i = 0; i < P; ++i) {
        PassArgs A;
        memset(&A, 0, sizeof A);
        A.L = L;
        A.s0 = s0;
        A.r = stages[i];
        A.first = i == 0;
        A.n_in = J.n_in;
        const bool last = i == P - 1;
        A.last = last;
        int colbits = A.first ? (L - A.r) : s0;
        static const int first_logT = [] { const char *e = ab_env("H2_NTT_LOGT_FIRST"); int v = e ? atoi(e) : -1; return v >= 0 && v <= 5 ? v : -1; }();   // sweeps only
        A.logT = std::min(A.first && first_logT >= 0 ? first_logT : want_logT, colbits);
        while (A.logT > 0 && ((32u << A.r) << A.logT) > lds_cap) A.logT--;
        // a transform alone on the chip, below 2^20: wide tiles are FEW tiles (2^18 as 10 + 8 stages at four columns = 64 workgroups
        // on 256 CUs) -- narrow them until there is one per CU.  Measured (profiles/r04_ntt_tile_width.txt): 2^19 0.0615 -> 0.0545 ms,
        // 2^18 0.0518 -> 0.0375, 2^17 0.0498 -> 0.0288, 2^16 0.0342 -> 0.0243.  Batched column transforms (plan 1) fill the chip
        // with columns instead and keep their 128-byte rows.
        if (J.plan == 0)
            while (A.logT > 0 && (n >> (A.r + A.logT)) < 256) A.logT--;
        // keep >= 256 lanes per workgroup when the pass is narrow
        while (A.logT < colbits && ((1 << A.r) << A.logT) < 1024 && ((32u << A.r) << (A.logT + 1)) <= lds_cap) A.logT++;
        if (A.first) {
            A.load_mode = J.load_mode;
            A.lk0 = to_param(J.lk0);
            A.lk1 = to_param(J.lk1);
        }
        if (last) {
            A.store_mode = J.store_mode;
            A.k0 = to_param(J.sk0);
            A.k1 = to_param(J.sk1);
            A.k2 = to_param(J.sk2);
        }
        // buffers: first pass is out of place (its store is a transposition); later passes in place;
        // the last pass lands in d_out
        const void *src;
        void *dst;
        if (P == 1) {
            src = J.d_in;
            dst = J.d_out;  // a single workgroup owns the whole vector: load-all then store-all
        } else if (J.d_in != J.d_out) {
            src = A.first ? J.d_in : J.d_out;
            dst = J.d_out;
        } else {
            src = A.first ? J.d_in : tmp;
            dst = last ? J.d_out : tmp;
        }
        // one lane per radix-4 group (tile / 4); a 1-stage pass needs tile / 2 butterflies, looped
        u32 threads = (u32)std::min<size_t>(1024, std::max<size_t>(64, ((size_t)1 << A.r << A.logT) / 4));
        size_t tiles = n >> (A.r + A.logT);
        size_t lds = ((size_t)32 << A.r) << A.logT;
        prof_begin(PROF_NTT_PASS, st);
        if ((rc = launch_pass(J.field, A, (unsigned)tiles, threads, lds, st, (const u32 *)src, (u32 *)dst, (const u32 *)tw->d)) != H2_OK)
            return rc;
        prof_end(PROF_NTT_PASS, st);
        s0 += A.r;
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

static bool bad_field(int field, int form) {
    return (field != H2_FP && field != H2_FQ) || (form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY);
}

static int job_ntt(NttJob &J, int field, void *d_a, unsigned log_n, const u64 *omega, int form) {
    memset(&J, 0, sizeof J);
    J.field = field;
    J.L = log_n;
    J.d_in = d_a;
    J.d_out = d_a;
    J.n_in = (size_t)1 << log_n;
    host_to_mont(field, J.omega, omega, form);
    return H2_OK;
}

}  // namespace h2

using namespace h2;

extern "C" int h2_ntt_device(int field, void *d_a, unsigned log_n, const uint64_t *omega, int form, void *stream) {
    if (bad_field(field, form) || !d_a || !omega || log_n > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    NttJob J;
    job_ntt(J, field, d_a, log_n, omega, form);
    return ntt_run(J, (hipStream_t)stream);
}

extern "C" int h2_ifft_device(int field, void *d_a, unsigned log_n, const uint64_t *omega_inv, const uint64_t *divisor,
                              int form, void *stream) {
    if (bad_field(field, form) || !d_a || !omega_inv || !divisor || log_n > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    NttJob J;
    job_ntt(J, field, d_a, log_n, omega_inv, form);
    J.store_mode = 1;
    host_to_mont(field, J.sk0, divisor, form);
    return ntt_run(J, (hipStream_t)stream);
}

// ---- independent column transforms in one call (plonk/prover.rs:111-117, 322-327): forked over internal streams with the
// small-tile plan so that they share the chip, joined on the caller's stream
namespace {
struct NttBatchStreams {
    std::mutex mu;
    std::vector<hipStream_t> s;
    std::vector<hipEvent_t> done;
    hipEvent_t fork = nullptr;
};
NttBatchStreams &ntt_batch_streams() {
    static NttBatchStreams b[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return b[dev & 15];
}
int ntt_batch(int field, void *const *d_a, size_t count, unsigned log_n, const uint64_t *omega, const uint64_t *divisor, int form,
              hipStream_t user) {
    NttBatchStreams &bs = ntt_batch_streams();
    std::lock_guard<std::mutex> lk(bs.mu);
    const size_t want = std::min<size_t>(3, count);
    while (bs.s.size() < want) {
        hipStream_t st;
        hipEvent_t ev;
        H2_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        H2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        bs.s.push_back(st);
        bs.done.push_back(ev);
    }
    if (!bs.fork) H2_HIP(hipEventCreateWithFlags(&bs.fork, hipEventDisableTiming));
    H2_HIP(hipEventRecord(bs.fork, user));
    for (size_t i = 0; i < want; ++i) H2_HIP(hipStreamWaitEvent(bs.s[i], bs.fork, 0));
    int rc = H2_OK;
    for (size_t i = 0; i < count && rc == H2_OK; ++i) {
        if (!d_a[i]) return H2_ERR_ARGS;
        NttJob J;
        job_ntt(J, field, d_a[i], log_n, omega, form);
        if (divisor) {
            J.store_mode = 1;
            host_to_mont(field, J.sk0, divisor, form);
        }
        J.plan = want > 1 ? 1 : 0;
        rc = ntt_run(J, bs.s[i % want]);
    }
    for (size_t i = 0; i < want; ++i) {
        H2_HIP(hipEventRecord(bs.done[i], bs.s[i]));
        H2_HIP(hipStreamWaitEvent(user, bs.done[i], 0));
    }
    return rc;
}
}  // namespace

extern "C" int h2_ntt_batch_device(int field, void *const *d_a, size_t count, unsigned log_n, const uint64_t *omega, int form, void *stream) {
    if (bad_field(field, form) || !d_a || !omega || log_n > 32) return H2_ERR_ARGS;
    if (!count) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return ntt_batch(field, d_a, count, log_n, omega, nullptr, form, (hipStream_t)stream);
}

extern "C" int h2_ifft_batch_device(int field, void *const *d_a, size_t count, unsigned log_n, const uint64_t *omega_inv,
                                    const uint64_t *divisor, int form, void *stream) {
    if (bad_field(field, form) || !d_a || !omega_inv || !divisor || log_n > 32) return H2_ERR_ARGS;
    if (!count) return H2_OK;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    return ntt_batch(field, d_a, count, log_n, omega_inv, divisor, form, (hipStream_t)stream);
}

extern "C" int h2_coeff_to_extended_device(int field, const void *d_a, void *d_out, unsigned k, unsigned ext_k,
                                           const uint64_t *g_coset, const uint64_t *g_coset_inv,
                                           const uint64_t *extended_omega, int form, void *stream) {
    if (bad_field(field, form) || !d_a || !d_out || d_a == d_out || !g_coset || !g_coset_inv || !extended_omega || ext_k > 32 ||
        k > ext_k)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    NttJob J;
    job_ntt(J, field, d_out, ext_k, extended_omega, form);
    J.d_in = d_a;
    J.n_in = (size_t)1 << k;
    J.load_mode = 1;
    host_to_mont(field, J.lk0, g_coset, form);
    host_to_mont(field, J.lk1, g_coset_inv, form);
    return ntt_run(J, (hipStream_t)stream);
}

extern "C" int h2_extended_to_coeff_device(int field, void *d_a, unsigned ext_k, const uint64_t *g_coset,
                                           const uint64_t *g_coset_inv, const uint64_t *extended_omega_inv,
                                           const uint64_t *extended_ifft_divisor, int form, void *stream) {
    if (bad_field(field, form) || !d_a || !g_coset || !g_coset_inv || !extended_omega_inv || !extended_ifft_divisor || ext_k > 32)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    NttJob J;
    job_ntt(J, field, d_a, ext_k, extended_omega_inv, form);
    J.store_mode = 2;
    // a[i] * divisor * {1, zeta^2, zeta}[i % 3]   (domain.rs:316, into_coset = false)
    u64 div[4], z[4], zi[4];
    host_to_mont(field, div, extended_ifft_divisor, form);
    host_to_mont(field, z, g_coset, form);
    host_to_mont(field, zi, g_coset_inv, form);
    memcpy(J.sk0, div, 32);
    host_mul(field, J.sk1, div, zi);
    host_mul(field, J.sk2, div, z);
    return ntt_run(J, (hipStream_t)stream);
}

// ---- divide_by_vanishing_poly (poly/domain.rs:329-348): a[i] *= t_evaluations[i mod nt] ----------------------------
template <int F>
__global__ void __launch_bounds__(256) k_mul_periodic(u32 *__restrict__ a, const u32 *__restrict__ t, size_t n, u32 nt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store(a + 8 * i, fe_mulx<F>(fe_load(a + 8 * i), fe_load(t + 8 * (size_t)(i % nt))));
}

extern "C" int h2_divide_by_vanishing_poly_device(int field, void *d_a, unsigned ext_k, const uint64_t *t_evaluations, size_t nt, int form,
                                                  void *stream) {
    if (bad_field(field, form) || !d_a || !t_evaluations || nt == 0 || nt > 4096 || ext_k > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    NttContext &cx = ntt_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    DevBuf &tb = cx.stage[std::make_pair(dev * 4 + 2, st)];
    if ((rc = tb.reserve(nt * 32)) != H2_OK) return rc;
    std::vector<u64> tm(nt * 4);
    for (size_t i = 0; i < nt; ++i) host_to_mont(field, &tm[4 * i], t_evaluations + 4 * i, form);   // Montgomery factors work for either data form
    H2_HIP(hipStreamSynchronize(st));   // the small table buffer is reused across calls
    H2_HIP(hipMemcpyAsync(tb.ptr, tm.data(), nt * 32, hipMemcpyHostToDevice, st));
    H2_HIP(hipStreamSynchronize(st));   // tm goes out of scope
    const size_t n = (size_t)1 << ext_k;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (field == H2_FP) hipLaunchKernelGGL((k_mul_periodic<FP>), grid, block, 0, st, (u32 *)d_a, tb.as<u32>(), n, (u32)nt);
    else hipLaunchKernelGGL((k_mul_periodic<FQ>), grid, block, 0, st, (u32 *)d_a, tb.as<u32>(), n, (u32)nt);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---- host-pointer variants: stage through a per-stream device buffer --------------------------------
namespace {
struct Staged {
    void *d = nullptr;
    int rc = H2_OK;
};
Staged stage_buffer(size_t bytes, int slot) {
    Staged s;
    NttContext &cx = ntt_ctx();
    std::lock_guard<std::mutex> lk(cx.mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    // slot-indexed staging buffers live beside the tmp buffers, keyed by a pseudo-stream
    DevBuf &b = cx.stage[std::make_pair(dev * 4 + slot, (hipStream_t) nullptr)];
    s.rc = b.reserve(bytes);
    s.d = b.ptr;
    return s;
}
std::mutex g_host_mu;  // host-pointer calls share staging buffers: serialise them
}  // namespace

extern "C" int h2_ntt(int field, uint64_t *a, unsigned log_n, const uint64_t *omega, int form) {
    if (bad_field(field, form) || !a || !omega || log_n > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t bytes = (size_t)32 << log_n;
    Staged s = stage_buffer(bytes, 0);
    if (s.rc != H2_OK) return s.rc;
    H2_HIP(hipMemcpyAsync(s.d, a, bytes, hipMemcpyHostToDevice, 0));
    if ((rc = h2_ntt_device(field, s.d, log_n, omega, form, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpyAsync(a, s.d, bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

extern "C" int h2_ifft(int field, uint64_t *a, unsigned log_n, const uint64_t *omega_inv, const uint64_t *divisor, int form) {
    if (bad_field(field, form) || !a || !omega_inv || !divisor || log_n > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t bytes = (size_t)32 << log_n;
    Staged s = stage_buffer(bytes, 0);
    if (s.rc != H2_OK) return s.rc;
    H2_HIP(hipMemcpyAsync(s.d, a, bytes, hipMemcpyHostToDevice, 0));
    if ((rc = h2_ifft_device(field, s.d, log_n, omega_inv, divisor, form, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpyAsync(a, s.d, bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

extern "C" int h2_coeff_to_extended(int field, const uint64_t *a, uint64_t *out, unsigned k, unsigned ext_k,
                                    const uint64_t *g_coset, const uint64_t *g_coset_inv, const uint64_t *extended_omega,
                                    int form) {
    if (bad_field(field, form) || !a || !out || !g_coset || !g_coset_inv || !extended_omega || ext_k > 32 || k > ext_k)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t in_bytes = (size_t)32 << k, out_bytes = (size_t)32 << ext_k;
    Staged si = stage_buffer(in_bytes, 1), so = stage_buffer(out_bytes, 0);
    if (si.rc != H2_OK) return si.rc;
    if (so.rc != H2_OK) return so.rc;
    H2_HIP(hipMemcpyAsync(si.d, a, in_bytes, hipMemcpyHostToDevice, 0));
    if ((rc = h2_coeff_to_extended_device(field, si.d, so.d, k, ext_k, g_coset, g_coset_inv, extended_omega, form, nullptr)) != H2_OK)
        return rc;
    H2_HIP(hipMemcpyAsync(out, so.d, out_bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

extern "C" int h2_divide_by_vanishing_poly(int field, uint64_t *a, unsigned ext_k, const uint64_t *t_evaluations, size_t nt, int form) {
    if (bad_field(field, form) || !a || !t_evaluations || nt == 0 || nt > 4096 || ext_k > 32) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t bytes = (size_t)32 << ext_k;
    Staged s = stage_buffer(bytes, 0);
    if (s.rc != H2_OK) return s.rc;
    H2_HIP(hipMemcpyAsync(s.d, a, bytes, hipMemcpyHostToDevice, 0));
    if ((rc = h2_divide_by_vanishing_poly_device(field, s.d, ext_k, t_evaluations, nt, form, nullptr)) != H2_OK) return rc;
    H2_HIP(hipMemcpyAsync(a, s.d, bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}

extern "C" int h2_extended_to_coeff(int field, uint64_t *a, unsigned ext_k, const uint64_t *g_coset,
                                    const uint64_t *g_coset_inv, const uint64_t *extended_omega_inv,
                                    const uint64_t *extended_ifft_divisor, int form) {
    if (bad_field(field, form) || !a || !g_coset || !g_coset_inv || !extended_omega_inv || !extended_ifft_divisor || ext_k > 32)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    std::lock_guard<std::mutex> lk(g_host_mu);
    size_t bytes = (size_t)32 << ext_k;
    Staged s = stage_buffer(bytes, 0);
    if (s.rc != H2_OK) return s.rc;
    H2_HIP(hipMemcpyAsync(s.d, a, bytes, hipMemcpyHostToDevice, 0));
    if ((rc = h2_extended_to_coeff_device(field, s.d, ext_k, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor, form,
                                          nullptr)) != H2_OK)
        return rc;
    H2_HIP(hipMemcpyAsync(a, s.d, bytes, hipMemcpyDeviceToHost, 0));
    H2_HIP(hipStreamSynchronize(0));
    return H2_OK;
}
