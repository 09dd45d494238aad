#include "msm_internal.cuh"

using namespace h2;

// ---- the paired commit over a SMALL 16-bit table: 8-bit sub-digits, 512 buckets, no bucket fold to speak of ----------------------
// The opening argument's rounds over the collapsed generators are paired commits of 2^14 .. 2^15 points, fifteen of them in a row at
// k = 20, each a chain of ~13 short launches through the machinery above: a two-pass sort into 2 x 2^15 buckets that hold eight
// entries each, an accumulate of four entries per lane, and a fold over 2^16 buckets (finish, two heavy-bucket launches that find
// nothing, 383 line sums per slice, 15 bit planes with a 15-doubling chain) -- 0.24 ms of which 0.05 is bucket arithmetic.  For a small
// table the same commit is cheaper with FEWER buckets: every signed 16-bit table digit d is cut once more, |d| = e_0 + 256 e_1 with e_0
// in [-127, 128] and e_1 in [0, 128] (the read-out of the collapsed generators does the same, ipa_readout_*), so that
//     sum_m c_m G_m = P_0 + 2^8 P_1,     P_pos = sum_{b < 128} (b + 1) * (sum of +-T[w][m] over the (m, w) whose sub-digit at `pos` is +-(b + 1))
// per output: 2 sides x 2 positions x 128 = 512 buckets in all, two entries per digit (2^20 entries for 2^15 + 4 scalars: the
// accumulate doubles, to the 50 us the chip needs for 2^20 mixed additions), a sort by a 9-bit key (three short launches, LDS
// histograms), a finisher in which EVERY bucket is a tree over ~256 range heads, 8 bit planes straight over each slice's 128 bucket sums
// (a slice is ONE line of the bucket matrix: no line sums) with a 7-doubling chain, and 8 doublings to join the positions.  The
// accumulate and the planes are the kernels above.  Bucket SLOTS lie 129 apart per slice (slot = key + key / 128: a slice's sums, then one
// slot that stays empty) -- the layout fold9_planes reads S column sums and NR - 1 row sums in, with S = 128 and NR = 1.
static constexpr u32 kSubKeys = 512;     // side (2) x position (2) x (|e| - 1 < 128)
static constexpr u32 kSubSlots = 516;    // 4 x 129
template <typename Fn> __device__ __forceinline__ void for_each_subdigit(const fe &s, u32 side, Fn f) {       // s: canonical, below 2^255
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const u32 raw = ((s.v[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;      // signed 16-bit digits, as msm_recode cuts them
        const bool neg = raw > 0x8000u;
        carry = neg ? 1u : 0u;
        const u32 mag = neg ? 0x10000u - raw : raw;                             // |d| <= 2^15 (raw = 2^16: digit 0, carry out)
        u32 e0 = mag & 255u, c8 = 0;
        bool e0neg = false;
        if (e0 > 128u) {
            e0 = 256u - e0;
            e0neg = true;
            c8 = 1;
        }
        const u32 e1 = (mag >> 8) + c8;                                         // <= 128: |d| <= 2^15, and |d| = 2^15 has e_0 = 0
        if (e0) f(side * 256u + e0 - 1u, (u32)w, (neg != e0neg) ? 0x80000000u : 0u);
        if (e1) f(side * 256u + 128u + e1 - 1u, (u32)w, neg ? 0x80000000u : 0u);
    }
    // (carry is 0 here: the scalar is below 2^255, the top window takes it)
}
// the same over an ENDOMORPHISM table (Bases::glv = 9): the scalar is split k = k1 + k2 lambda (glv.cuh, |k1|, |k2| < 2^129), each half cut into nine
// signed 16-bit digits -- the ninth holds bit 128 and the last carry -- and every digit into its two sub-digits; the digits of k1 read rows 0 .. 8,
// those of k2 rows 9 .. 17 (the images under phi); a negative half flips the sign of all its entries.  f(key, ROW, sign).
template <int FS, typename Fn> __device__ __forceinline__ void for_each_subdigit_glv(const fe &s, u32 side, Fn f) {
    u32 mag[2][5], hneg[2];
    glv_split<FS>(s, mag[0], hneg[0], mag[1], hneg[1]);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const bool hn = hneg[part] != 0;
        u32 carry = 0;
#pragma unroll
        for (int w = 0; w < 7; ++w) {                                                // windows 0 .. 6: signed, as in for_each_subdigit
            const u32 raw = ((mag[part][w >> 1] >> (16 * (w & 1))) & 0xFFFFu) + carry;
            const bool neg = raw > 0x8000u;
            carry = neg ? 1u : 0u;
            const u32 m = neg ? 0x10000u - raw : raw;
            u32 e0 = m & 255u, c8 = 0;
            bool e0neg = false;
            if (e0 > 128u) {
                e0 = 256u - e0;
                e0neg = true;
                c8 = 1;
            }
            const u32 e1 = (m >> 8) + c8;
            const bool dn = neg != hn;                                                // the digit's sign times the half's
            if (e0) f(side * 256u + e0 - 1u, (u32)(part * 9 + w), (dn != e0neg) ? 0x80000000u : 0u);
            if (e1) f(side * 256u + 128u + e1 - 1u, (u32)(part * 9 + w), dn ? 0x80000000u : 0u);
        }
        // Window 7 is cut UNSIGNED (0 .. 2^16 with the carry): recoded like the others it would send a carry into window 8 for a quarter of
        // the halves, every one of those entries into the SAME bucket (position 0, magnitude 1) -- four times the average bucket, and the
        // finisher's launch is as long as its longest tree.  Its high sub-digit may then exceed 128 (up to 257): it leaves as two or three
        // entries of at most 128 each, cut evenly so that they spread over the buckets of position 1.
        {
            const u32 raw = (mag[part][3] >> 16) + carry;
            u32 e0 = raw & 255u, c8 = 0;
            bool e0neg = false;
            if (e0 > 128u) {
                e0 = 256u - e0;
                e0neg = true;
                c8 = 1;
            }
            u32 e1 = (raw >> 8) + c8;
            if (e0) f(side * 256u + e0 - 1u, (u32)(part * 9 + 7), (hn != e0neg) ? 0x80000000u : 0u);
            // (split EVENLY: "128 and the rest" would pile a quarter of the halves into the one bucket of magnitude 128)
            const u32 pieces = (e1 + 127u) / 128u;                  // 0 .. 3
            for (u32 j = 0; j < pieces; ++j) {
                const u32 d = (e1 + j) / pieces;                    // floor((e1 + j) / pieces), j < pieces: sums to e1, each <= 128
                f(side * 256u + 128u + d - 1u, (u32)(part * 9 + 7), hn ? 0x80000000u : 0u);
            }
        }
        // window 8: bit 128 and beyond (glv_split promises < 2^129; nothing has been seen above 2^128) -- unsigned too, almost always nothing
        {
            u32 rest = min(mag[part][4], 256u);        // (<= 1 by glv_split's bound, which tests/test_glv_constants.py holds the constants to; the clamp keeps a
                                                       // broken promise from running past the entry list: two entries per half are reserved for this window)
            while (rest) {
                const u32 d = rest > 128u ? 128u : rest;
                rest -= d;
                f(side * 256u + d - 1u, (u32)(part * 9 + 8), hn ? 0x80000000u : 0u);
            }
        }
    }
}
__device__ __forceinline__ u32 pair_side(u32 i, u32 pair_n, int pair_shift) { return i < pair_n ? (i >> pair_shift) & 1u : (i - pair_n) & 1u; }
// pass A: a workgroup's 512 scalars -> its 512 counters (H2_SUB_BLOCK = 256 / 512 / 1024, late round of the k = 20 argument: 0.220 / 0.219 / 0.227 ms --
// fewer workgroups shorten pass B's walk, more of them pass C's scattered stores)
static constexpr u32 kSubBlock = 512;
template <int FS>
__global__ void __launch_bounds__(1024) sub_count(const u32 *__restrict__ scalars, u32 n, u32 pair_n, int pair_shift, int mont, int glv,
                                                       u32 *__restrict__ wg_hist) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kSubKeys];
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        fe s = fe_load(scalars + 8 * (size_t)i);
        if (mont) s = fe_redc<FS>(s);
        if (glv) for_each_subdigit_glv<FS>(s, pair_side(i, pair_n, pair_shift), [&](u32 key, u32, u32) { atomicAdd(&sh[key], 1u); });
        else for_each_subdigit(s, pair_side(i, pair_n, pair_shift), [&](u32 key, u32, u32) { atomicAdd(&sh[key], 1u); });
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) wg_hist[(size_t)blockIdx.x * kSubKeys + k] = sh[k];
}
// pass B (one workgroup, a lane per key): the workgroups' offsets inside each key's run, every key's start (`kstart`, for pass C), the boundary
// array over the 516 SLOTS (+ total + the sentinel msm_accumulate reads; a gap slot is an empty bucket), and the raw bucket slots cleared for the
// accumulate (36 words each: a lane clears its own, lanes 0 .. 3 the gaps too)
__global__ void __launch_bounds__(kSubKeys) sub_scan(const u32 *__restrict__ wg_hist, u32 nblk, u32 *__restrict__ wg_off, u32 *__restrict__ kstart,
                                                     u32 *__restrict__ starts, u32 *__restrict__ buckets9) {
    H2_LATENCY_STAGE();
    __shared__ u32 tot[kSubKeys];
    const u32 k = threadIdx.x;
    u32 run = 0;
    for (u32 b0 = 0; b0 < nblk; b0 += 8) {               // eight loads in flight: the walk is a chain of memory round trips otherwise
        u32 c[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) c[j] = b0 + j < nblk ? wg_hist[(size_t)(b0 + j) * kSubKeys + k] : 0u;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
            if (b0 + j < nblk) wg_off[(size_t)(b0 + j) * kSubKeys + k] = run;
            run += c[j];
        }
    }
    const u32 slot = k + (k >> 7);
#pragma unroll
    for (int i = 0; i < 36; ++i) buckets9[36 * (size_t)slot + i] = 0u;
    if (k < 4)
        for (int i = 0; i < 36; ++i) buckets9[36 * (size_t)(129 * k + 128) + i] = 0u;
    tot[k] = run;
    __syncthreads();
    for (u32 off = 1; off < kSubKeys; off <<= 1) {
        const u32 t = k >= off ? tot[k - off] : 0u;
        __syncthreads();
        tot[k] += t;
        __syncthreads();
    }
    kstart[k] = tot[k] - run;
    starts[slot] = tot[k] - run;
    if ((k & 127u) == 127u) starts[slot + 1] = tot[k];          // the gap behind a slice: starts where the next slice starts
    if (k == kSubKeys - 1) {
        starts[kSubSlots] = tot[k];
        starts[kSubSlots + 1] = 0xFFFFFFFFu;
    }
}
// pass C: the same digits again, each to its place (entry = table index | sign << 31; the order inside a bucket is immaterial)
template <int FS>
__global__ void __launch_bounds__(1024) sub_scatter(const u32 *__restrict__ scalars, u32 n, u32 pair_n, int pair_shift, int mont, int glv, u32 stride,
                                                   const u32 *__restrict__ wg_off, const u32 *__restrict__ kstart, u32 *__restrict__ entries) {
    H2_LATENCY_STAGE();
    __shared__ u32 cur[kSubKeys];
    for (u32 k = threadIdx.x; k < kSubKeys; k += blockDim.x) cur[k] = kstart[k] + wg_off[(size_t)blockIdx.x * kSubKeys + k];
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe s = fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_redc<FS>(s);
    auto place = [&](u32 key, u32 row, u32 sign) {
        const u32 pos = atomicAdd(&cur[key], 1u);
        entries[pos] = (row * stride + i) | sign;
    };
    if (glv) for_each_subdigit_glv<FS>(s, pair_side(i, pair_n, pair_shift), place);
    else for_each_subdigit(s, pair_side(i, pair_n, pair_shift), place);
}
// the finisher when EVERY bucket owns hundreds of range heads: a workgroup per bucket, a tree over its heads, then the bucket's own segment
template <int FB>
__global__ void __launch_bounds__(256, 3) fold9_finish_dense(const u32 *__restrict__ heads9, const u32 *__restrict__ starts, u32 *__restrict__ buckets9,
                                                           u32 total_buckets, u32 T, u32 div) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    const u32 b = blockIdx.x, base = starts[0];
    const u32 M = starts[total_buckets] - base;
    T = eff_lanes(M, T, div);
    const u32 chunk = max(1u, (M + T - 1) / T);
    const u32 h0 = (starts[b] - base + chunk - 1) / chunk, h1 = (starts[b + 1] - base + chunk - 1) / chunk;
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(heads9, h1 > h0 ? h1 - h0 : 0u, [h0](u32 k) { return h0 + k; });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (!fold9_root()) return;
    xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(buckets9 + 36 * (size_t)b));
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz9_store_raw<FB>(buckets9 + 36 * (size_t)b, acc);
}
// The rest of the fold of the four 128-bucket slices in ONE launch (fold9_planes' scheme, without line sums and with the join of the two
// positions inside): workgroup (t, y) sums plane t of slice y = side * 2 + pos -- the 64 finished buckets with bit t of b + 1 set (plane 7:
// bucket 127 alone) -- and doubles it t times; the workgroup that arrives LAST at its slice adds the eight planes (three tree levels), doubles
// the sum eight more times when the slice is a position 1 (its buckets count in units of 2^8), and of the two slices of a side the one that
// arrives last adds the other's sum and writes output `side`.  Arrival counters behind fences, left at zero: counter[0..3] the slices',
// counter[4..5] the sides'.
template <int FB>
__global__ void __launch_bounds__(256, 3) sub_planes(const u32 *__restrict__ buckets9, u32 *__restrict__ planes9, u32 *__restrict__ sums9,
                                                   u32 *__restrict__ counter, u32 *__restrict__ out, int out_kind, int out_mont) {
    H2_LATENCY_STAGE();
    __shared__ __attribute__((aligned(16))) u32 sh[32 * 36];
    __shared__ u32 s_last;
    const u32 t = blockIdx.x, y = blockIdx.y;
    const bool lead = (threadIdx.x & (kGroup - 1)) == 0;
    const u32 *src = buckets9 + 36 * (size_t)129 * y;
    planes9 += 36 * (size_t)8 * y;
    xyzz9<FB> acc = fold9_quad_gather<FB, H2_FOLD_D>(src, t < 7 ? 64u : 1u, [t](u32 k) {
        return t < 7 ? ((((k >> t) << (t + 1)) | (1u << t) | (k & ((1u << t) - 1u))) - 1u) : 127u;      // the k-th b with bit t of b + 1 set
    });
    acc = fold9_quads_sum<FB>(acc, sh);
    if (fold9_root()) {
        for (u32 k = 0; k < t; ++k) acc = xyzz9_dbl_wide<FB>(acc);
        if (lead) {
            xyzz9_store_raw<FB>(planes9 + 36 * (size_t)t, acc);
            __threadfence();
            s_last = atomicAdd(counter + y, 1u) == 7u ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    acc = fold9_quad_gather<FB, H2_FOLD_D>(planes9, 8u, [](u32 k) { return k; });
    acc = fold9_quads_sum<FB>(acc, sh, 8u);
    if (!fold9_root()) return;
    if (y & 1u)
        for (int k = 0; k < 8; ++k) acc = xyzz9_dbl_wide<FB>(acc);
    u32 ticket = 0;
    if (lead) {
        counter[y] = 0;
        xyzz9_store_raw<FB>(sums9 + 36 * (size_t)y, acc);
        __threadfence();
        ticket = atomicAdd(counter + 4 + (y >> 1), 1u);
    }
    ticket = (u32)__builtin_amdgcn_mov_dpp((int)ticket, 0, 0xf, 0xf, false);      // quad lane 0's ticket
    if (ticket == 0) return;                                                       // the side's other slice finishes the output
    __threadfence();
    xyzz9_add_wide<FB>(acc, xyzz9_load_raw<FB>(sums9 + 36 * (size_t)(y ^ 1u)));
    const xyzz<FB> r = xyzz9_to_r256_wide<FB>(acc);
    if (!lead) return;
    counter[4 + (y >> 1)] = 0;
    out += (out_kind == H2_OUT_AFFINE ? 16 : 24) * (size_t)(y >> 1);
    if (out_kind == H2_OUT_AFFINE) {
        affine<FB> a = xyzz_to_affine<FB>(r);
        if (!out_mont) { a.x = fe_from_mont<FB>(a.x); a.y = fe_from_mont<FB>(a.y); }
        fe_store(out, a.x);
        fe_store(out + 8, a.y);
    } else {
        fe X, Y, Z;
        xyzz_to_jacobian<FB>(r, X, Y, Z);
        if (!out_mont) { X = fe_from_mont<FB>(X); Y = fe_from_mont<FB>(Y); Z = fe_from_mont<FB>(Z); }
        fe_store(out, X);
        fe_store(out + 8, Y);
        fe_store(out + 16, Z);
    }
}

template <int FB, int FS>
static int pair_subdigit_launch(MsmContext &cx, const Bases &b, const void *d_scalars, size_t n, unsigned pair_shift, int form, int out_kind,
                                void *d_out, hipStream_t st) {
    int rc;
    static const u32 sub_block = [] { const char *e = ab_env("H2_SUB_BLOCK"); int v = e ? atoi(e) : 0; return (u32)(v == 256 || v == 512 || v == 1024 ? v : kSubBlock); }();   // A/B
    const u32 nblk = (u32)((n + sub_block - 1) / sub_block), tb = kSubSlots, nsl = 4;
    const int glv = b.glv ? 1 : 0;
    const size_t max_entries = n * (glv ? 40 : 32);             // two sub-digits per table digit: 16 digits; over an endomorphism table 2 x (7 x 2 + 4 + 2) at most
    u32 &lanes = cx.lanes[FB][2];
    if (!lanes) {            // how many lanes of the M9 accumulate the chip holds at once (as msm_launch sizes it)
        int dev = 0, cus = 0, per_cu = 0;
        H2_HIP(hipGetDevice(&dev));
        H2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        H2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)msm_accumulate<FB, false, true>, 256, 0));
        per_cu = std::min(per_cu, (int)H2_ACC9_WAVES);
        lanes = (u32)cus * (u32)std::max(per_cu, 1) * 256u;
    }
    // entries per lane: 8 over a plain table (2^14 + 4 scalars: 2^19 entries, 128 range heads per bucket -- exactly what the finisher's 64 quads take in
    // ONE round of two gathers each); an endomorphism table leaves ~33 entries per scalar, which at 8 per lane is 131 heads per bucket and a second
    // round for three of them (finisher 35 -> 54 us): 9 per lane there
    const u32 lane_div = glv ? 9 : 8;
    const u32 T = (u32)std::min<size_t>(lanes, std::max<size_t>(256, (max_entries / lane_div + 255) / 256 * 256));
    if ((rc = cx.hist.reserve(((size_t)2 * nblk + 1) * kSubKeys * 4)) != H2_OK || (rc = cx.starts.reserve((tb + 2) * 4)) != H2_OK ||
        (rc = cx.entries.reserve(max_entries * 4)) != H2_OK || (rc = cx.seg9.reserve(((size_t)T + tb) * 144)) != H2_OK ||
        (rc = cx.partial.reserve((size_t)nsl * (8 + 1) * 144)) != H2_OK)
        return rc;
    if (cx.fold_ctr.cap < 64) {          // fold9_planes' arrival counters: zero once, every launch leaves them at zero
        if ((rc = cx.fold_ctr.reserve((size_t)kMaxCols * 64)) != H2_OK) return rc;
        H2_HIP(hipMemsetAsync(cx.fold_ctr.ptr, 0, (size_t)kMaxCols * 64, st));
    }
    u32 *wg_hist = cx.hist.as<u32>(), *wg_off = wg_hist + (size_t)nblk * kSubKeys, *kstart = wg_off + (size_t)nblk * kSubKeys;
    u32 *starts = cx.starts.as<u32>(), *entries = cx.entries.as<u32>();
    u32 *heads9 = cx.seg9.as<u32>(), *buckets9 = heads9 + 36 * (size_t)T;
    u32 *planes9 = cx.partial.as<u32>(), *sums9 = planes9 + 36 * (size_t)nsl * 8;
    const int mont = form == H2_FORM_MONTGOMERY ? 1 : 0;
    const u32 pair_n = (u32)(n - 4);
    ColStride cs;
    memset(&cs, 0, sizeof cs);
    hipLaunchKernelGGL((sub_count<FS>), dim3(nblk), dim3(sub_block), 0, st, (const u32 *)d_scalars, (u32)n, pair_n, (int)pair_shift, mont, glv, wg_hist);
    hipLaunchKernelGGL(sub_scan, dim3(1), dim3(kSubKeys), 0, st, (const u32 *)wg_hist, nblk, wg_off, kstart, starts, buckets9);
    hipLaunchKernelGGL((sub_scatter<FS>), dim3(nblk), dim3(sub_block), 0, st, (const u32 *)d_scalars, (u32)n, pair_n, (int)pair_shift, mont, glv, b.stride,
                       (const u32 *)wg_off, (const u32 *)kstart, entries);
    hipLaunchKernelGGL((msm_accumulate<FB, false, true>), dim3(T / 256), dim3(256), 0, st, (const u32 *)b.d_table, (const u32 *)nullptr, 0xFFFFFFFFu,
                       (const u32 *)entries, (const u32 *)starts, heads9, buckets9, tb, T, lane_div, cs);
    hipLaunchKernelGGL((fold9_finish_dense<FB>), dim3(tb), dim3(256), 0, st, (const u32 *)heads9, (const u32 *)starts, buckets9, tb, T, lane_div);
    hipLaunchKernelGGL((sub_planes<FB>), dim3(8, nsl), dim3(256), 0, st, (const u32 *)buckets9, planes9, sums9, cx.fold_ctr.as<u32>(), (u32 *)d_out, out_kind,
                       mont);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

extern "C" int h2_commit_pair_device(h2_bases_t g, const void *d_scalars, size_t n, unsigned pair_shift, int form, int out_kind,
                                     void *d_out, void *stream) {
    auto b = find_bases(g, true);
    if (!b) return H2_ERR_HANDLE;
    if (bad_common(b->curve, form, out_kind) || !d_out || !d_scalars || n != b->n || n < 8 || pair_shift > 31) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    // small 16-bit tables (the opening argument's rounds over the collapsed generators): the 8-bit sub-digit form above.
    // H2_PAIR_SUBDIGITS=0: the general form for every size (A/B); = n: the largest table (points) that takes the sub-digit form.
    if (b->glv || (pair_subdigits_apply(n) && b->c == 16 && b->W == 16 && !prof_enabled() && !timeline_on())) {      // (an endomorphism table has no other reader)
        if (b->curve == H2_PALLAS) return pair_subdigit_launch<FP, FQ>(cx, *b, d_scalars, n, pair_shift, form, out_kind, d_out, st);
        return pair_subdigit_launch<FQ, FP>(cx, *b, d_scalars, n, pair_shift, form, out_kind, d_out, st);
    }
    MsmArgs a{d_scalars, nullptr, b->d_table, nullptr, n, true, b->c, b->stride, (u32)b->n, form, out_kind, d_out};
    a.pair_shift = (int)pair_shift;
    a.pair_n = (u32)(n - 4);
    return msm_dispatch(cx, b->curve, a, st);
}

