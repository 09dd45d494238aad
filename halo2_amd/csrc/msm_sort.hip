// Bucket sort of the window digits of a multiexp (see msm_launch.hip for the stages of the whole): the one-pass counting sort of small
// problems (recode -> count -> scan -> scatter) and the two-pass sort of large ones (pass 1 by the top bucket bits, pass 2 a workgroup per bin).
#include "msm_internal.cuh"

namespace h2 {

// ---- recode: scalars -> signed window digits -------------------------------------------------
// code = 0xFFFF for digit 0, else (|d| - 1) | (d < 0 ? 0x8000 : 0);  digits[w * m + i]
template <int FS>
__global__ void __launch_bounds__(256) msm_recode(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont) {
    H2_LATENCY_STAGE();
    // m counts the optional extra (blind) scalar, which is column m - 1 and lives in its own buffer
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fe s = (extra_scalar && i == m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_from_mont<FS>(s);
    u32 carry = 0;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
    for (int w = 0; w < W; ++w) {
        int bit = w * c, word = bit >> 5, sh = bit & 31;
        u64 two = (u64)limb_at(s, word) | ((u64)limb_at(s, word + 1) << 32);  // limb_at(.., 8) = 0
        u32 raw = ((u32)(two >> sh) & mask) + carry;
        u32 code;
        if (raw > half) {
            carry = 1;
            code = ((1u << c) - raw - 1) | 0x8000u;   // d = raw - 2^c <= 0: |d| - 1 (d = 0 wraps to 0xFFFF)
        } else {
            carry = 0;
            code = raw ? raw - 1 : kZeroCode;
        }
        digits[(size_t)w * m + i] = (uint16_t)code;
    }
}

// ---- recode with the endomorphism split (generic path): scalar i yields two columns of signed digits, column i for k1
// (base P_i) and column m + i for k2 (base phi(P_i)); half as many windows, so the final Horner over windows needs 128
// doublings instead of 255.  digits[w * 2m + col]
template <int FS>
__global__ void __launch_bounds__(256) msm_recode_glv(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont) {
    H2_LATENCY_STAGE();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fe s = fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_from_mont<FS>(s);
    u32 mag[2][5], neg[2];
    glv_split<FS>(s, mag[0], neg[0], mag[1], neg[1]);
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
    const size_t M = 2 * (size_t)m;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 carry = 0;
        for (int w = 0; w < W; ++w) {
            const int bit = w * c, word = bit >> 5, sh = bit & 31;
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                lo = (word == q) ? mag[part][q] : lo;
                hi = (word + 1 == q) ? mag[part][q] : hi;
            }
            const u64 two = (u64)lo | ((u64)hi << 32);
            const u32 raw = ((u32)(two >> sh) & mask) + carry;
            // digits of |k_part| in [-half + 1, half] (or [-half, half - 1] when the part is negative, so that after the
            // sign flip every digit is again in the encodable set: the code has no room for -half)
            const bool up = neg[part] ? raw >= half : raw > half;
            carry = up;
            const u32 digit_mag = up ? (1u << c) - raw : raw;     // 0 when raw = 0 or raw = 2^c
            const u32 negative = (up ? 1u : 0u) ^ neg[part];
            const u32 code = digit_mag ? ((digit_mag - 1) | (negative ? 0x8000u : 0u)) : kZeroCode;
            digits[(size_t)w * M + (size_t)part * m + i] = (uint16_t)code;
        }
    }
}

// ---- count: LDS histogram per (slice, chunk) ---------------------------------------------------
__global__ void __launch_bounds__(1024) msm_count(const uint16_t *__restrict__ digits, u32 *__restrict__ hist,
                                                  size_t items, u32 chunk, u32 NB) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 h[];
    const u32 b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) h[j] = 0;
    __syncthreads();
    size_t lo = (size_t)b * chunk, hi = min(items, lo + chunk);
    const uint16_t *d = digits + (size_t)sl * items;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) atomicAdd(&h[code & 0x7FFFu], 1u);
    }
    __syncthreads();
    u32 *dst = hist + ((size_t)sl * B + b) * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) dst[j] = h[j];
}

// ---- scan a: per-bucket totals, chunk slices become exclusive prefixes -------------------------
__global__ void __launch_bounds__(256) msm_chunk_prefix(u32 *__restrict__ hist, u32 *__restrict__ counts, u32 NB,
                                                        u32 B, u32 total_buckets) {
    H2_LATENCY_STAGE();
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_buckets) return;
    u32 sl = g / NB, j = g % NB;
    u32 run = 0;
    for (u32 b = 0; b < B; ++b) {
        size_t k = ((size_t)sl * B + b) * NB + j;
        u32 t = hist[k];
        hist[k] = run;
        run += t;
    }
    counts[g] = run;
}

// ---- scan b: exclusive scan of the bucket totals -> entry offsets (starts[total] = M) ----------------
// three kernels: block sums, scan of block sums, apply.
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *sh, u32 &total) {
    // blockDim.x == kScanBlock
    const u32 t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (u32 off = 1; off < kScanBlock; off <<= 1) {
        u32 x = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    total = sh[kScanBlock - 1];
    u32 r = sh[t] - v;
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_blocksums(const u32 *__restrict__ counts, u32 *__restrict__ bsums,
                                                                 u32 total) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 g = blockIdx.x * kScanBlock + threadIdx.x;
    u32 tot;
    (void)block_excl_scan(g < total ? counts[g] : 0, sh, tot);
    if (threadIdx.x == 0) bsums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_top(u32 *__restrict__ bsums, u32 nblocks, u32 *__restrict__ grand) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 carry = 0;
    for (u32 base = 0; base < nblocks; base += kScanBlock) {
        u32 i = base + threadIdx.x;
        u32 v = i < nblocks ? bsums[i] : 0, tot;
        u32 ex = block_excl_scan(v, sh, tot);
        if (i < nblocks) bsums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand = carry;
}
__global__ void __launch_bounds__(kScanBlock) msm_scan_apply(const u32 *__restrict__ counts, const u32 *__restrict__ bsums,
                                                             const u32 *__restrict__ grand, u32 *__restrict__ starts,
                                                             u32 total) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    u32 g = blockIdx.x * kScanBlock + threadIdx.x;
    u32 tot;
    u32 a = block_excl_scan(g < total ? counts[g] : 0, sh, tot) + bsums[blockIdx.x];
    if (g < total) starts[g] = a;
    if (g == 0) {
        starts[total] = *grand;
        starts[total + 1] = 0xFFFFFFFFu;      // sentinel: msm_accumulate reads the boundary after next without a bounds check
    }
}

// ---- scatter: bucket-sorted entry list ----------------------------------------------------------
// entry = base index | sign << 31.  generic: base index = column (column m-1 of a blinded commit maps to
// `extra_col`); registered: base index = w * stride + column, straight into the precomputed table.
__global__ void __launch_bounds__(1024) msm_scatter(const uint16_t *__restrict__ digits, const u32 *__restrict__ hist,
                                                    const u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    size_t items, u32 chunk, u32 NB, u32 m, u32 stride, u32 extra_col,
                                                    int table, u32 col0) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 off[];
    const u32 b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
    const u32 *src = hist + ((size_t)sl * B + b) * NB;
    const u32 *st = starts + (size_t)sl * NB;
    for (u32 j = threadIdx.x; j < NB; j += blockDim.x) off[j] = st[j] + src[j];
    __syncthreads();
    size_t lo = (size_t)b * chunk, hi = min(items, lo + chunk);
    const uint16_t *d = digits + (size_t)sl * items;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        u32 code = d[i];
        if (code != kZeroCode) {
            u32 pos = atomicAdd(&off[code & 0x7FFFu], 1u);
            const u32 i32 = (u32)i;  // items < 2^31
            u32 w = table ? i32 / m : 0, col = table ? i32 % m : i32;
            if (col == m - 1 && extra_col != 0xFFFFFFFFu) col = extra_col;
            else col += col0;
            entries[pos] = (w * stride + col) | ((code & 0x8000u) << 16);
        }
    }
}

// signed window digits of one scalar (same recoding as msm_recode, 32-bit codes so that windows may exceed 16 bits),
// handed to f(w, code): code = kZero32 for digit 0, else (|d| - 1) | (d < 0) << 31
template <int C, typename Fn> __device__ __forceinline__ void for_each_digit_static(const fe &s, Fn f) {
    constexpr int W = 255 / C + 1;
    constexpr u32 mask = (1u << C) - 1, half = 1u << (C - 1), full = 1u << C;
    u32 carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        constexpr int dummy = 0;
        (void)dummy;
        const int bit = w * C, word = bit >> 5, sh = bit & 31;     // compile-time after unrolling: plain register picks
        u32 v = s.v[word] >> sh;
        if (sh + C > 32 && word + 1 < 8) v |= s.v[word + 1] << (32 - sh);
        const u32 raw = (v & mask) + carry;
        carry = raw > half;
        const u32 code = carry ? (raw == full ? kZero32 : ((full - raw - 1) | 0x80000000u)) : (raw ? raw - 1 : kZero32);   // raw = 2^C: digit 0, carry out
        f(w, code);
    }
}
template <typename Fn> __device__ __forceinline__ void for_each_digit(const fe &s, int c, int W, Fn f) {
    // compile-time widths: the limb picks become plain register selects (the generic loop below indexes the limbs dynamically: at
    // c = 17 the pass-1 kernels took 29 + 78 us against 16 + 53 us for the unrolled c = 20)
    if (c == 16) { for_each_digit_static<16>(s, f); return; }
    if (c == 17) { for_each_digit_static<17>(s, f); return; }
    if (c == 18) { for_each_digit_static<18>(s, f); return; }
    if (c == 19) { for_each_digit_static<19>(s, f); return; }
    if (c == 20) { for_each_digit_static<20>(s, f); return; }
    if (c == 13) { for_each_digit_static<13>(s, f); return; }
    u32 carry = 0;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1), full = 1u << c;
    for (int w = 0; w < W; ++w) {
        int bit = w * c, word = bit >> 5, sh = bit & 31;
        u64 two = (u64)limb_at(s, word) | ((u64)limb_at(s, word + 1) << 32);
        const u32 raw = ((u32)(two >> sh) & mask) + carry;
        carry = raw > half;
        const u32 code = carry ? (raw == full ? kZero32 : ((full - raw - 1) | 0x80000000u)) : (raw ? raw - 1 : kZero32);
        f(w, code);
    }
}

// every non-zero window digit of scalar i as f(sort key, entry base index, sign << 31).
// Registered path: key = bucket (one slice), base = w * stride + column.  Generic path (GLV): the scalar is split
// (glv.cuh), key = w * nb + bucket (slice-major), base = digit column (i for k1 / P_i, m + i for k2 / phi(P_i)).
template <int FS, bool GLV, typename Fn>
__device__ __forceinline__ void emit_entries(const fe &s, u32 i, const Sort2 &P, Fn f) {
    if (!GLV) {
        u32 col = P.col0 + i;
        if (i == P.m - 1 && P.extra_col != 0xFFFFFFFFu) col = P.extra_col;
        u32 side_key = 0;
        if (P.pair_shift >= 0) side_key = (i < P.pair_n ? (i >> P.pair_shift) & 1u : (i - P.pair_n) & 1u) * P.nb;
        for_each_digit(s, P.c, P.W, [&](int w, u32 code) {
            if (code != kZero32) f(side_key + (code & 0x7FFFFFFFu), (u32)w * P.stride + col, code & 0x80000000u, (u32)w);
        });
        return;
    }
    u32 mag[2][5], neg[2];
    glv_split<FS>(s, mag[0], neg[0], mag[1], neg[1]);
    const int c = P.c;
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 carry = 0;
        for (int w = 0; w < P.W; ++w) {
            const int bit = w * c, word = bit >> 5, sh = bit & 31;
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                lo = (word == q) ? mag[part][q] : lo;
                hi = (word + 1 == q) ? mag[part][q] : hi;
            }
            const u32 raw = ((u32)(((u64)lo | ((u64)hi << 32)) >> sh) & mask) + carry;
            const bool up = neg[part] ? raw >= half : raw > half;      // same digit set as msm_recode_glv
            carry = up;
            const u32 digit_mag = up ? (1u << c) - raw : raw;
            if (digit_mag) f((u32)w * P.nb + digit_mag - 1, (u32)part * P.m + i, (((up ? 1u : 0u) ^ neg[part]) << 31), (u32)w);
        }
    }
}

// pass 1, COUNT: hist1[blk][h] = this workgroup's entries per bin
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_count(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [nh] counters
    if (gridDim.z > 1) {
        scalars = ci.scalars[blockIdx.z];
        extra_scalar = ci.blinds[blockIdx.z];
        hist1 = H2_COLZ(hist1, cs.hist);
    }
    const u32 nh = P.nh, blk = blockIdx.x;
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) sh[h] = 0;
    __syncthreads();
    for (u32 loc = threadIdx.x; loc < P.s1_scalars; loc += blockDim.x) {
        const u32 i = blk * P.s1_scalars + loc;
        if (i >= P.m) break;
        fe s = (extra_scalar && i == P.m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
        if (P.mont) s = fe_redc<FS>(s);
        emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32, u32, u32) { atomicAdd(&sh[key >> P.lowb], 1u); });
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) hist1[(size_t)blk * nh + h] = sh[h];
}

// pass 1, SCATTER: the workgroup's entries are first grouped by bin in LDS (its per-bin counts are known from the
// count pass), then every bin's run goes out as one contiguous copy -- scattered 4-byte stores issue one lane per
// clock and were the cost of this pass.  hist1 holds the exclusive prefix over workgroups by now.
template <int FS, bool GLV>
__global__ void __launch_bounds__(1024) msm_s1_scatter(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    if (gridDim.z > 1) {
        scalars = ci.scalars[blockIdx.z];
        extra_scalar = ci.blinds[blockIdx.z];
        hist1 = H2_COLZ(hist1, cs.hist);
        bin_count = H2_COLZ(bin_count, cs.plan);
        bin_start = H2_COLZ(bin_start, cs.plan);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
    }
    const u32 nh = P.nh, blk = blockIdx.x, B1 = gridDim.x;
    u32 *gstart = sh;                 // [nh] bin_start, then bin_start + this workgroup's offset inside the bin
    u32 *lstart = sh + nh;            // [nh + 1] where the bin's run begins in the stage
    u32 *cursor = lstart + nh + 1;    // [nh]
    u32 *stage = cursor + nh;         // [s1_scalars * digits per scalar] entries
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        const u32 mine = hist1[(size_t)blk * nh + h];
        const u32 next = blk + 1 < B1 ? hist1[(size_t)(blk + 1) * nh + h] : bin_count[h];
        gstart[h] = bin_count[h];
        lstart[h] = next - mine;      // this workgroup's entries in bin h
        cursor[h] = mine;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 M = wave0_excl_scan(gstart, nh);
        const u32 L = wave0_excl_scan(lstart, nh);
        if (threadIdx.x == 0) {
            lstart[nh] = L;
            if (blk == 0) bin_start[nh] = M;
        }
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        if (blk == 0) bin_start[h] = gstart[h];
        gstart[h] += cursor[h];
        cursor[h] = lstart[h];
    }
    __syncthreads();
    const u32 lowmask = (1u << P.lowb) - 1;
    for (u32 loc = threadIdx.x; loc < P.s1_scalars; loc += blockDim.x) {
        const u32 i = blk * P.s1_scalars + loc;
        if (i >= P.m) break;
        fe s = (extra_scalar && i == P.m - 1) ? fe_load(extra_scalar) : fe_load(scalars + 8 * (size_t)i);
        if (P.mont) s = fe_redc<FS>(s);
        if (!GLV && P.side) {
            // the stage word keeps what the copy-out needs to rebuild the entry: scalar (11 bits, s1_scalars <= 2048), window
            // (6 bits) and the low bucket bits (<= 14), so entry and low bits leave as two contiguous runs per bin
            emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32, u32 sign, u32 w) {
                const u32 pos = atomicAdd(&cursor[key >> P.lowb], 1u);
                stage[pos] = loc | (w << 11) | ((key & lowmask) << 17) | sign;
            });
        } else {
            emit_entries<FS, GLV>(s, i, P, [&](u32 key, u32 base, u32 sign, u32) {
                const u32 pos = atomicAdd(&cursor[key >> P.lowb], 1u);
                stage[pos] = base | ((key & lowmask) << P.lb) | sign;
            });
        }
    }
    __syncthreads();
    // P.run_lanes (16) lanes per bin at a time: contiguous LDS run -> contiguous global run.  A workgroup's run of a bin is ~30
    // entries at 2048 scalars x 15 digits over 1024 bins and a bin costs its group three dependent LDS reads before the first
    // store, whatever the run's length: the loop is as long as the bins a group walks (whole waves: 2x; half waves -> 16 lanes: -10 us)
    const u32 kRunLanes = P.run_lanes, wave = threadIdx.x / kRunLanes, lane = threadIdx.x & (kRunLanes - 1), nwaves = blockDim.x / kRunLanes;
    if (!GLV && P.side) {
        for (u32 h = wave; h < nh; h += nwaves) {
            const u32 l0 = lstart[h], l1 = lstart[h + 1];
            u32 *dst = tagged + gstart[h];
            uint16_t *dlo = tagged_low + gstart[h];
            for (u32 q = l0 + lane; q < l1; q += kRunLanes) {
                const u32 word = stage[q], i = blk * P.s1_scalars + (word & 2047u), w = (word >> 11) & 63u;
                u32 col = P.col0 + i;
                if (i == P.m - 1 && P.extra_col != 0xFFFFFFFFu) col = P.extra_col;
                dst[q - l0] = (w * P.stride + col) | (word & 0x80000000u);
                dlo[q - l0] = (uint16_t)((word >> 17) & 0x3FFFu);
            }
        }
        return;
    }
    for (u32 h = wave; h < nh; h += nwaves) {
        const u32 l0 = lstart[h], l1 = lstart[h + 1];
        u32 *dst = tagged + gstart[h];
        for (u32 q = l0 + lane; q < l1; q += kRunLanes) dst[q - l0] = stage[q];
    }
}

// column-wise exclusive scan of hist1[B1][nh]; bin_count[h] = column total.  16 columns per workgroup, 64 row groups.
// (It also zeroes the two small counter areas later kernels of the same multiexp count into -- `z2`: the two words of the
// heavy-bucket list, `z1`: the oversized-bin counter of pass 2 -- which saves two 5 us memset nodes on the stream.)
__global__ void __launch_bounds__(1024) msm_s1_prefix(u32 *__restrict__ hist1, u32 *__restrict__ bin_count, u32 B1, u32 nh, u32 *__restrict__ z2,
                                                      u32 *__restrict__ z1, u32 *__restrict__ sentinel, ColStride cs) {
    H2_LATENCY_STAGE();
    __shared__ u32 part[64][17];
    if (gridDim.z > 1) {
        hist1 = H2_COLZ(hist1, cs.hist);
        bin_count = H2_COLZ(bin_count, cs.plan);
        z2 = H2_COLZ(z2, cs.heavy);
        if (z1) z1 = H2_COLZ(z1, cs.plan);
        sentinel = H2_COLZ(sentinel, cs.starts);
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        if (threadIdx.x < 2) z2[threadIdx.x] = 0;
        else if (threadIdx.x == 2) { if (z1) z1[0] = 0; }
        else *sentinel = 0xFFFFFFFFu;       // starts[total_buckets + 1]: what msm_accumulate reads past the last boundary -- written HERE for
                                            // every form of pass 2 (one-launch bins, oversized bins, chunked); the one-pass sort: msm_scan_apply
    }
    const u32 r = threadIdx.x >> 4, cl = threadIdx.x & 15, col = blockIdx.x * 16 + cl;
    const u32 rg = (B1 + 63) / 64, r0 = min(B1, r * rg), r1 = min(B1, r0 + rg);
    u32 sum = 0;
    if (col < nh)
        for (u32 row = r0; row < r1; ++row) sum += hist1[(size_t)row * nh + col];
    part[r][cl] = sum;
    __syncthreads();
    u32 run = 0;
    for (u32 q = 0; q < r; ++q) run += part[q][cl];
    if (col < nh) {
        for (u32 row = r0; row < r1; ++row) {
            const size_t k = (size_t)row * nh + col;
            const u32 t = hist1[k];
            hist1[k] = run;
            run += t;
        }
        if (r == 63) bin_count[col] = run;
    }
}

// largest h in [0, nh) with bin_start[h] <= p   (bin_start non-decreasing, bin_start[0] = 0 <= p < bin_start[nh])
__device__ __forceinline__ u32 bin_of(const u32 *__restrict__ bin_start, u32 nh, u32 p) {
    u32 lo = 0, hi = nh;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (bin_start[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

// pass-2 plan: first bin and histogram offset of every chunk (window = the bins the chunk touches); one workgroup
__global__ void __launch_bounds__(kScanBlock) msm_s2_plan(const u32 *__restrict__ bin_start, Sort2 P, u32 *__restrict__ hlo,
                                                          u32 *__restrict__ woff) {
    H2_LATENCY_STAGE();
    __shared__ u32 sh[kScanBlock];
    const u32 M = bin_start[P.nh];
    u32 carry = 0;
    for (u32 base = 0; base < P.B2; base += kScanBlock) {
        const u32 cidx = base + threadIdx.x;
        u32 size = 0, h0 = 0;
        if (cidx < P.B2 && (size_t)cidx * P.K2 < M) {
            const u32 p0 = cidx * P.K2, p1 = min(M, p0 + P.K2) - 1;
            h0 = bin_of(bin_start, P.nh, p0);
            size = (bin_of(bin_start, P.nh, p1) - h0 + 1) << P.lowb;
        }
        u32 tot;
        const u32 ex = block_excl_scan(size, sh, tot);
        if (cidx < P.B2) {
            hlo[cidx] = h0;
            woff[cidx] = carry + ex;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) woff[P.B2] = carry;
}

// bin (relative to the chunk's first bin) of list position p; bounds[q] = bin_start[h0 + q + 1]
__device__ __forceinline__ u32 rel_bin(const u32 *bounds, u32 nbins, u32 p) {
    if (p < bounds[0]) return 0;
    u32 lo = 0, hi = nbins - 1;                                 // invariant: bounds[lo] <= p < bounds[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (bounds[mid] <= p) lo = mid;
        else hi = mid;
    }
    return hi;
}

// low bucket bits of tagged entry p: in the entry's spare bits, or in the side array
__device__ __forceinline__ u32 s2_low(const Sort2 &P, const uint16_t *__restrict__ low, u32 p, u32 e, u32 lowmask) {
    return P.side ? (u32)low[p] : (e >> P.lb) & lowmask;
}

// pass 2, COUNT over one chunk of the tagged list: hist2[woff[c] + (bucket - window base)]
__global__ void __launch_bounds__(1024) msm_s2_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                     const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ hist2) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [window] counters, then the window's bin boundaries
    const u32 cidx = blockIdx.x, M = bin_start[P.nh];
    const size_t p0 = (size_t)cidx * P.K2;
    if (p0 >= M) return;
    const u32 p1 = (u32)min((size_t)M, p0 + P.K2);
    const u32 h0 = hlo[cidx], wo = woff[cidx], wsize = woff[cidx + 1] - wo, nbins = wsize >> P.lowb;
    const u32 lowmask = (1u << P.lowb) - 1;
    if (wsize > P.lds_window) {
        // a very sparse column: the chunk's window does not fit LDS.  Few entries by construction -- count in HBM
        // (hist2 is zeroed before this kernel).
        const u32 *gb = bin_start + h0 + 1;
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            atomicAdd(&hist2[wo + ((rel_bin(gb, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask))], 1u);
        }
        return;
    }
    u32 *bounds = sh + wsize;
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) sh[k] = 0u;
    for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];
        atomicAdd(&sh[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) hist2[wo + k] = sh[k];
}

// pass 2, SCATTER: final entries (low bits stripped) at starts[bucket] + the chunk's share (hist2 holds the exclusive
// prefix over chunks by now).  Windows of up to kS2StageWindow buckets -- every chunk of a dense or moderately sparse
// column -- group the chunk by bucket in LDS first and copy each bucket's run out contiguously; wider windows write
// straight from the counters.
__global__ void __launch_bounds__(1024) msm_s2_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                       const u32 *__restrict__ hlo, const u32 *__restrict__ woff, Sort2 P,
                                                       u32 *__restrict__ hist2, const u32 *__restrict__ starts, u32 *__restrict__ entries) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 cidx = blockIdx.x, M = bin_start[P.nh];
    const size_t p0 = (size_t)cidx * P.K2;
    if (p0 >= M) return;
    const u32 p1 = (u32)min((size_t)M, p0 + P.K2);
    const u32 h0 = hlo[cidx], wo = woff[cidx], wsize = woff[cidx + 1] - wo, nbins = wsize >> P.lowb;
    const u32 lowmask = (1u << P.lowb) - 1, strip = P.side ? ~0u : ~(lowmask << P.lb);
    if (wsize > P.lds_window) {             // very sparse column: hist2 (exclusive offsets by now) doubles as the cursor
        const u32 *gb = bin_start + h0 + 1;
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            const u32 k = (rel_bin(gb, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask);
            entries[starts[(h0 << P.lowb) + k] + atomicAdd(&hist2[wo + k], 1u)] = e & strip;
        }
        return;
    }
    if (wsize > kS2StageWindow) {
        u32 *bounds = sh + wsize;
        for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) sh[k] = starts[(h0 << P.lowb) + k] + hist2[wo + k];
        for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
        __syncthreads();
        for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const u32 e = tagged[p];
            const u32 pos = atomicAdd(&sh[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
            entries[pos] = e & strip;
        }
        return;
    }
    u32 *gstart = sh;                      // [wsize] where the chunk's run of bucket k starts in `entries`
    u32 *lstart = gstart + wsize;          // [wsize + 1] ... and in the stage
    u32 *cursor = lstart + wsize + 1;      // [wsize]
    u32 *bounds = cursor + wsize;          // [nbins]
    u32 *stage = bounds + nbins;           // [K2] entries grouped by bucket
    uint16_t *kid = reinterpret_cast<uint16_t *>(stage + P.K2);   // [K2] window-relative bucket of each staged entry
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) {
        gstart[k] = starts[(h0 << P.lowb) + k] + hist2[wo + k];
        lstart[k] = 0;
    }
    for (u32 q = threadIdx.x; q < nbins; q += blockDim.x) bounds[q] = bin_start[h0 + q + 1];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];
        atomicAdd(&lstart[(rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 L = wave0_excl_scan(lstart, wsize);
        if (threadIdx.x == 0) lstart[wsize] = L;
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < wsize; k += blockDim.x) cursor[k] = lstart[k];
    __syncthreads();
    for (u32 p = (u32)p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const u32 e = tagged[p];                                 // second read of the chunk comes from L2
        const u32 k = (rel_bin(bounds, nbins, p) << P.lowb) | s2_low(P, tagged_low, p, e, lowmask);
        const u32 pos = atomicAdd(&cursor[k], 1u);
        stage[pos] = e & strip;
        kid[pos] = (uint16_t)k;
    }
    __syncthreads();
    // consecutive lanes copy consecutive staged entries: runs of one bucket leave as contiguous stores
    for (u32 q = threadIdx.x; q < p1 - (u32)p0; q += blockDim.x) {
        const u32 k = kid[q];
        entries[gstart[k] + (q - lstart[k])] = stage[q];
    }
}

// ---- pass 2 in ONE launch: a workgroup per pass-1 bin ----------------------------------------------------------------------
// The chunked pass 2 above cuts the tagged list into 16 K-entry chunks that may straddle bins, so it needs a plan, a count
// pass, a per-bucket prefix over the chunks, a three-kernel scan of all bucket totals and then the scatter: seven launches,
// the tagged list read twice from HBM.  But pass 1 already knows where every bin begins (bin_start), a bin is one contiguous
// run of the tagged list, and for anything but a pathological column it fits in LDS (2^20 scalars, 17-bit windows: 1024 bins
// of ~15 K entries).  So: one workgroup per bin counts its 2^lowb buckets in LDS, scans them -- starts[bucket] = bin_start +
// local prefix, no global scan -- groups the bin by bucket in LDS and writes it out as ONE contiguous copy.  The tagged list is
// read once from HBM (the second read of a bin hits L2), `entries` is written in full lines.  A bin that does not fit (tens of
// thousands of equal scalars) is scattered straight to memory by the same workgroup.
// Counters are bumped with a wave-aggregated form: when every active lane of a wave holds the same key (a column of equal
// scalars) one lane adds the population count instead of 64 lanes serialising on one LDS address.
__device__ __forceinline__ u32 lds_ticket(u32 *ctr, u32 k) {
    const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)k);
    const unsigned long long act = __ballot(1), same = __ballot(k == k0);
    if (same == act) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0u));
        u32 base = 0;
        if (rank == 0) base = atomicAdd(&ctr[k0], (u32)__popcll(act));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        return base + rank;
    }
    return atomicAdd(&ctr[k], 1u);
}
// the same for four keys per lane: when all four are valid and every active lane's four keys are ONE key, a single atomic takes
// 4 x population; else four tickets.  pos[j] is only written for j < nvalid.
__device__ __forceinline__ void lds_ticket4(u32 *ctr, const u32 k[4], u32 nvalid, u32 pos[4]) {
    const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)k[0]);
    const bool mine = nvalid == 4 && k[0] == k0 && k[1] == k0 && k[2] == k0 && k[3] == k0;
    const unsigned long long act = __ballot(1), same = __ballot(mine);
    if (same == act) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0u));
        u32 base = 0;
        if (rank == 0) base = atomicAdd(&ctr[k0], 4u * (u32)__popcll(act));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base) + 4u * rank;
#pragma unroll
        for (int j = 0; j < 4; ++j) pos[j] = base + j;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((u32)j < nvalid) pos[j] = lds_ticket(ctr, k[j]);
}
__global__ void __launch_bounds__(1024) msm_s2_bins(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                    Sort2 P, u32 total_buckets, u32 cap, u32 *__restrict__ starts, u32 *__restrict__ entries,
                                                    u32 *__restrict__ big, u32 max_big, u32 *__restrict__ zero9, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    u32 cb = 0;                                                         // joined columns: where this column's entries begin
    if (gridDim.z > 1) {
        if (zero9) zero9 = H2_COLZ(zero9, cs.buckets);
        cb = col_entry_base(bin_start, cs);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        starts = H2_COLZ(starts, cs.starts);
        entries = H2_COLZ(entries, cs.entries);
        big = H2_COLZ(big, cs.plan);
    }
    const u32 nbk = 1u << P.lowb, h = blockIdx.x;
    u32 *cnt = sh, *cursor = sh + nbk, *stage = cursor + nbk;          // [nbk] | [nbk] | [cap]
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], E = p1 - p0, q0 = cb + p0;     // q0: the bin's place in the sorted list
    const u32 lowmask = nbk - 1, strip = P.side ? ~0u : ~(lowmask << P.lb);
    // M  (the sentinel behind it: msm_s1_prefix).  Joined columns: that slot is bucket 0 of the next column, which writes the same value.
    if (h == gridDim.x - 1 && threadIdx.x == 0) starts[total_buckets] = cb + bin_start[gridDim.x];
    if (zero9) {
        // the raw bucket slots msm_accumulate parks segments in start from zero: this bin's buckets are cleared HERE (a memset node
        // less on the stream of every commit; the slots are not touched again before the accumulate)
        const u32 b0 = h << P.lowb, b1 = min(total_buckets, b0 + nbk);
        if (b1 > b0) {
            uint4 *z = reinterpret_cast<uint4 *>(zero9 + 36 * (size_t)b0);
            for (u32 i = threadIdx.x; i < 9 * (b1 - b0); i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
        }
    }
    if (E > cap && max_big) {
        // a bin that does not fit the stage (a degenerate column: every scalar equal, half of them 1 ...) goes on the list of big
        // bins, which msm_s2_big_* sort with kBigChunks workgroups each; only past kMaxBig such bins does this workgroup do it alone
        if (threadIdx.x == 0) {
            const u32 slot = atomicAdd(&big[0], 1u);
            if (slot < max_big) big[1 + slot] = h;
            cnt[0] = slot;
        }
        __syncthreads();
        const u32 slot = cnt[0];
        __syncthreads();
        if (slot < max_big) return;
    }
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    // four consecutive entries per lane and trip: four loads in flight per lane instead of one (a bin of a degenerate column --
    // every scalar equal, or half of them 1 -- holds up to 2^20 entries and is streamed by this one workgroup, twice)
    const u32 step = blockDim.x * 4;
    for (u32 base = p0 + threadIdx.x * 4; base < p1; base += step) {
        u32 e[4], k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < p1 ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < p1 ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        u32 pos[4];
        lds_ticket4(cnt, k, min(4u, p1 - base), pos);
    }
    __syncthreads();
    if (threadIdx.x < 64) (void)wave0_excl_scan(cnt, nbk);              // cnt[k] = entries of the bin before bucket k
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        const u32 b = (h << P.lowb) + k;
        cursor[k] = cnt[k];
        if (b < total_buckets) starts[b] = q0 + cnt[k];
    }
    __syncthreads();
    const bool fits = E <= cap;
    for (u32 base = p0 + threadIdx.x * 4; base < p1; base += step) {    // second read of the bin: L2 (a bin beyond the stage that found no
                                                                        // slot on the big-bin list: scattered by this workgroup alone)
        u32 e[4], k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < p1 ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < p1 ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        u32 pos[4];
        lds_ticket4(cursor, k, min(4u, p1 - base), pos);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < p1) {
                if (fits) stage[pos[j]] = e[j] & strip;
                else entries[q0 + pos[j]] = e[j] & strip;               // a bin beyond the stage: scattered straight to memory
            }
    }
    if (!fits) return;
    __syncthreads();
    for (u32 i = threadIdx.x; i < E; i += blockDim.x) entries[q0 + i] = stage[i];
}

// ---- big bins (listed by msm_s2_bins): kBigChunks workgroups per bin -- count, prefix, scatter.  Three small launches that
// return at once when the list is empty (the common case: ~14 us), so that a degenerate column costs what it cost with the
// chunked pass 2 instead of being streamed by one workgroup per bin (every scalar equal: 1.68 ms against 0.88).
__global__ void __launch_bounds__(1024) msm_s2_big_count(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                         Sort2 P, const u32 *__restrict__ big, u32 *__restrict__ gcnt, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 s = blockIdx.y, c = blockIdx.x;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, lowmask = nbk - 1, h = big[1 + s];
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], csize = (p1 - p0 + kBigChunks - 1) / kBigChunks;
    const u32 a = min(p1, p0 + c * csize), b = min(p1, a + csize);
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    for (u32 base = a + threadIdx.x * 4; base < b; base += blockDim.x * 4) {
        u32 e[4], k[4], pos[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < b ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < b ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        lds_ticket4(sh, k, min(4u, b - base), pos);
    }
    __syncthreads();
    u32 *dst = gcnt + ((size_t)s * kBigChunks + c) * nbk;
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) dst[k] = sh[k];
}
__global__ void __launch_bounds__(1024) msm_s2_big_prefix(const u32 *__restrict__ bin_start, Sort2 P, u32 total_buckets, const u32 *__restrict__ big,
                                                          u32 *__restrict__ gcnt, u32 *__restrict__ starts, ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];      // [nbk] bucket totals -> exclusive prefix
    const u32 s = blockIdx.x;
    u32 cb = 0;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        cb = col_entry_base(bin_start, cs);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
        starts = H2_COLZ(starts, cs.starts);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, h = big[1 + s], p0 = cb + bin_start[h];
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        u32 run = 0;
        for (u32 c = 0; c < kBigChunks; ++c) {
            u32 *q = gcnt + ((size_t)s * kBigChunks + c) * nbk + k;
            const u32 t = *q;
            *q = run;                                               // this chunk's offset inside bucket k
            run += t;
        }
        sh[k] = run;
    }
    __syncthreads();
    if (threadIdx.x < 64) (void)wave0_excl_scan(sh, nbk);
    __syncthreads();
    u32 *base = gcnt + (size_t)kMaxBig * kBigChunks * nbk + (size_t)s * nbk;       // bucket offsets inside the bin, for the scatter
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) {
        const u32 b = (h << P.lowb) + k;
        base[k] = sh[k];
        if (b < total_buckets) starts[b] = p0 + sh[k];
    }
}
__global__ void __launch_bounds__(1024) msm_s2_big_scatter(const u32 *__restrict__ tagged, const uint16_t *__restrict__ tagged_low, const u32 *__restrict__ bin_start,
                                                           Sort2 P, const u32 *__restrict__ big, const u32 *__restrict__ gcnt, u32 *__restrict__ entries,
                                                           ColStride cs) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 s = blockIdx.y, c = blockIdx.x;
    u32 cb = 0;
    if (gridDim.z > 1) {
        big = H2_COLZ(big, cs.plan);
        if (s >= min(big[0], kMaxBig)) return;
        cb = col_entry_base(bin_start, cs);
        tagged = H2_COLZ(tagged, cs.items);
        if (tagged_low) tagged_low = H2_COLZ(tagged_low, cs.items);
        bin_start = H2_COLZ(bin_start, cs.plan);
        gcnt = H2_COLZ(gcnt, cs.plan);
        entries = H2_COLZ(entries, cs.entries);
    }
    if (s >= min(big[0], kMaxBig)) return;
    const u32 nbk = 1u << P.lowb, lowmask = nbk - 1, strip = P.side ? ~0u : ~(lowmask << P.lb), h = big[1 + s];
    const u32 p0 = bin_start[h], p1 = bin_start[h + 1], csize = (p1 - p0 + kBigChunks - 1) / kBigChunks;
    const u32 a = min(p1, p0 + c * csize), b = min(p1, a + csize);
    const u32 *off = gcnt + ((size_t)s * kBigChunks + c) * nbk, *base_k = gcnt + (size_t)kMaxBig * kBigChunks * nbk + (size_t)s * nbk;
    for (u32 k = threadIdx.x; k < nbk; k += blockDim.x) sh[k] = base_k[k] + off[k];
    __syncthreads();
    for (u32 base = a + threadIdx.x * 4; base < b; base += blockDim.x * 4) {
        u32 e[4], k[4], pos[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = base + j < b ? tagged[base + j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = base + j < b ? s2_low(P, tagged_low, base + j, e[j], lowmask) : 0u;
        lds_ticket4(sh, k, min(4u, b - base), pos);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < b) entries[cb + p0 + pos[j]] = e[j] & strip;
    }
}

// per-bucket totals; each chunk's count becomes the bucket-relative offset of that chunk
__global__ void __launch_bounds__(256) msm_s2_prefix(u32 *__restrict__ hist2, const u32 *__restrict__ bin_start, const u32 *__restrict__ hlo,
                                                     const u32 *__restrict__ woff, Sort2 P, u32 *__restrict__ counts, u32 NB) {
    H2_LATENCY_STAGE();
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= NB) return;
    const u32 h = j >> P.lowb, b0 = bin_start[h], b1 = bin_start[h + 1];
    u32 run = 0;
    if (b1 > b0) {
        const u32 c0 = b0 / P.K2, c1 = (b1 - 1) / P.K2;
        for (u32 cidx = c0; cidx <= c1; ++cidx) {
            const size_t k = (size_t)woff[cidx] + (j - (hlo[cidx] << P.lowb));
            const u32 t = hist2[k];
            hist2[k] = run;
            run += t;
        }
    }
    counts[j] = run;
}


// ---- the GROUPED sort of a large generic multiexp (msm_generic.hip) -----------------------------------------------------------------
// best_multiexp without a registered table (arithmetic.rs:143-180): every scalar is split by the endomorphism into two half-length
// digit columns.  The sort of round 5 ran over all window slices at once and did that split in BOTH of its passes (count and scatter),
// in 72-register kernels that do not fit beside the accumulate of another call; at 2^21 / 2^22 points its bins outgrew LDS and pass 2
// fell back to the chunked seven-launch form (sort 0.59 / 1.37 ms of a 2.8 / 5.6 ms call).  Now the split happens ONCE
// (msm_glv_digits: a row of 16-bit digit codes per window), and the window slices are sorted in GROUPS, upper slices first: a group's
// pass 1 reads only its rows (2 bytes per entry, coalesced), its bins fit LDS at every size, and the groups after the first are sorted
// on a side stream while the first is being accumulated -- the sort in front of the first addition is a third of what it was.
//
// digits[w * row + col]: col = i for k1 / P_i, m + i for k2 / phi(P_i); row = 2 m rounded up to 8 (pass 1 reads eight codes a load).
template <int FS>
__global__ void __launch_bounds__(256) msm_glv_digits(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W,
                                                      int mont) {
    H2_LATENCY_STAGE();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fe s = fe_load(scalars + 8 * (size_t)i);
    if (mont) s = fe_redc<FS>(s);
    u32 mag[2][5], neg[2];
    glv_split<FS>(s, mag[0], neg[0], mag[1], neg[1]);
    const u32 mask = (1u << c) - 1, half = 1u << (c - 1);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        u32 carry = 0;
        for (int w = 0; w < W; ++w) {
            const int bit = w * c, word = bit >> 5, sh = bit & 31;
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                lo = (word == q) ? mag[part][q] : lo;
                hi = (word + 1 == q) ? mag[part][q] : hi;
            }
            const u32 raw = ((u32)(((u64)lo | ((u64)hi << 32)) >> sh) & mask) + carry;
            const bool up = neg[part] ? raw >= half : raw > half;      // same digit set as msm_recode_glv / emit_entries
            carry = up;
            const u32 digit_mag = up ? (1u << c) - raw : raw;
            const u32 negative = (up ? 1u : 0u) ^ neg[part];
            digits[(size_t)w * row + (size_t)part * m + i] = (uint16_t)(digit_mag ? ((digit_mag - 1) | (negative ? 0x8000u : 0u)) : kZeroCode);
        }
    }
}

// pass 1 of a group, COUNT: workgroup `blk` owns digit columns [blk S, (blk + 1) S) of the group's ns rows; hist1[blk][h] = its entries per bin
// (key = (w - w0) nb + bucket over the group's buckets, bin = key >> lowb)
__global__ void __launch_bounds__(512) msm_d1_count(const uint16_t *__restrict__ digits, GroupSort P, u32 *__restrict__ hist1) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [nh] counters
    const u32 nh = P.nh, blk = blockIdx.x;
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) sh[h] = 0;
    __syncthreads();
    const u32 c0 = blk * P.S, c1 = min(P.cols, c0 + P.S), vecs = (c1 - c0 + 7) / 8;
    for (u32 v = threadIdx.x; v < vecs * P.ns; v += blockDim.x) {
        const u32 s = v / vecs, col = c0 + 8 * (v - s * vecs);
        const uint4 d = *reinterpret_cast<const uint4 *>(digits + (size_t)(P.w0 + s) * P.row + col);
        const u32 wd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 code = (wd[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            if (col + j < c1 && code != kZeroCode) atomicAdd(&sh[(s * P.nb + (code & 0x7FFFu)) >> P.lowb], 1u);
        }
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) hist1[(size_t)blk * nh + h] = sh[h];
}

// pass 1 of a group, SCATTER: as msm_s1_scatter -- the workgroup's entries are grouped by bin in LDS, every bin's run leaves as one
// contiguous copy -- with the entries read from the digit rows.  tagged word = column | low bucket bits << lb | sign << 31.
__global__ void __launch_bounds__(512) msm_d1_scatter(const uint16_t *__restrict__ digits, GroupSort P, const u32 *__restrict__ hist1,
                                                      const u32 *__restrict__ bin_count, u32 *__restrict__ bin_start, u32 *__restrict__ tagged) {
    H2_LATENCY_STAGE();
    extern __shared__ __attribute__((aligned(16))) u32 sh[];
    const u32 nh = P.nh, blk = blockIdx.x, B1 = gridDim.x;
    u32 *gstart = sh;                 // [nh] bin_start, then bin_start + this workgroup's offset inside the bin
    u32 *lstart = sh + nh;            // [nh + 1] where the bin's run begins in the stage
    u32 *cursor = lstart + nh + 1;    // [nh]
    u32 *stage = cursor + nh;         // [S * ns] entries
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        const u32 mine = hist1[(size_t)blk * nh + h];
        const u32 next = blk + 1 < B1 ? hist1[(size_t)(blk + 1) * nh + h] : bin_count[h];
        gstart[h] = bin_count[h];
        lstart[h] = next - mine;      // this workgroup's entries in bin h
        cursor[h] = mine;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 M = wave0_excl_scan(gstart, nh);
        const u32 L = wave0_excl_scan(lstart, nh);
        if (threadIdx.x == 0) {
            lstart[nh] = L;
            if (blk == 0) bin_start[nh] = M;
        }
    }
    __syncthreads();
    for (u32 h = threadIdx.x; h < nh; h += blockDim.x) {
        if (blk == 0) bin_start[h] = gstart[h];
        gstart[h] += cursor[h];
        cursor[h] = lstart[h];
    }
    __syncthreads();
    const u32 lowmask = (1u << P.lowb) - 1;
    const u32 c0 = blk * P.S, c1 = min(P.cols, c0 + P.S), vecs = (c1 - c0 + 7) / 8;
    for (u32 v = threadIdx.x; v < vecs * P.ns; v += blockDim.x) {
        const u32 s = v / vecs, col = c0 + 8 * (v - s * vecs);
        const uint4 d = *reinterpret_cast<const uint4 *>(digits + (size_t)(P.w0 + s) * P.row + col);
        const u32 wd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 code = (wd[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            if (col + j < c1 && code != kZeroCode) {
                const u32 key = s * P.nb + (code & 0x7FFFu);
                const u32 pos = atomicAdd(&cursor[key >> P.lowb], 1u);
                stage[pos] = (col + j) | ((key & lowmask) << P.lb) | ((code & 0x8000u) << 16);
            }
        }
    }
    __syncthreads();
    const u32 kRunLanes = P.run_lanes, grp = threadIdx.x / kRunLanes, lane = threadIdx.x & (kRunLanes - 1), ngrp = blockDim.x / kRunLanes;
    for (u32 h = grp; h < nh; h += ngrp) {
        const u32 l0 = lstart[h], l1 = lstart[h + 1];
        u32 *dst = tagged + gstart[h];
        for (u32 q = l0 + lane; q < l1; q += kRunLanes) dst[q - l0] = stage[q];
    }
}

// ---- explicit instantiations (both curves): the host side lives in msm_launch.hip / msm_generic.hip ----
template __global__ void msm_recode<FP>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont);
template __global__ void msm_recode<FQ>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar,
                                                  uint16_t *__restrict__ digits, u32 m, int c, int W, int mont);
template __global__ void msm_recode_glv<FP>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont);
template __global__ void msm_recode_glv<FQ>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, int c, int W,
                                                      int mont);
template __global__ void msm_s1_count<FP, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
template __global__ void msm_s1_count<FP, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
template __global__ void msm_s1_count<FQ, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
template __global__ void msm_s1_count<FQ, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                     u32 *__restrict__ hist1, ColIn ci, ColStride cs);
template __global__ void msm_s1_scatter<FP, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
template __global__ void msm_s1_scatter<FP, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
template __global__ void msm_s1_scatter<FQ, false>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);
template __global__ void msm_s1_scatter<FQ, true>(const u32 *__restrict__ scalars, const u32 *__restrict__ extra_scalar, Sort2 P,
                                                       const u32 *__restrict__ hist1, const u32 *__restrict__ bin_count,
                                                       u32 *__restrict__ bin_start, u32 *__restrict__ tagged, uint16_t *__restrict__ tagged_low,
                                                       ColIn ci, ColStride cs);

template __global__ void msm_glv_digits<FP>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W, int mont);
template __global__ void msm_glv_digits<FQ>(const u32 *__restrict__ scalars, uint16_t *__restrict__ digits, u32 m, u32 row, int c, int W, int mont);

}  // namespace h2
