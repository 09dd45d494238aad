// The collapsed generators of the opening argument read straight out of a registered table (poly/commitment/prover.rs:154-166).
#include "msm_internal.cuh"

namespace h2 {

// ---- the collapsed generators of the opening argument, straight from a registered table ------------------------------------
// After J rounds G'_J[i] = sum_{h < 2^J} s(h) * G[i + h * nJ] (nJ = 2^(k-J); s(h) = the challenge products of
// h2_ipa_round_scalars_device), i.e. nJ multiexps of 2^J terms that SHARE their scalars.  Each s(h) is cut into the table's
// 16-bit signed digits d_w and every d_w into four signed 4-bit digits e_v in [-7, 8]:
//     s(h) G[m] = sum_w sum_v 16^v e_{h,w,v} T[w][m],          T[w][m] = 2^(16w) G[m] (the table's row w)
// so bucket (v, b) of output i collects +-T[w][i + h nJ] over the (h, w) with |e_{h,w,v}| = b -- the SAME (h, w, sign) list for
// every i.  The host writes the 32 lists once (2^J * 64 entries in all); lane i of workgroup row (v, b) walks list (v, b) with
// no divergence and perfectly coalesced 64-byte gathers (consecutive lanes read consecutive table columns), 2^J * 64 mixed
// additions per output in the carry-free field layer.  ipa_collapse_windows then forms sum_b b * bucket per (i, v) by running
// sums and ipa_collapse_finish the Horner step over v (12 doublings) and the affine result.
template <int FB>
__global__ void __launch_bounds__(256, H2_ACC9_WAVES) ipa_collapse_buckets(const u32 *__restrict__ table, const u32 *__restrict__ list,
                                                                         const u32 *__restrict__ list_start, u32 nJ, u32 *__restrict__ sums) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lb = blockIdx.y;
    if (i >= nJ) return;
    const u32 lo = list_start[lb], hi = list_start[lb + 1];
    xyzz9<FB> acc = xyzz9_identity<FB>();
    if (lo < hi) {
        // loads unconditional (clamped at the tail), as in msm_accumulate: a conditional load makes hipcc wait for everything in flight
        u32 e0 = list[lo], e1 = list[min(lo + 1, hi - 1)];
        affine<FB> nxt = aff_load<FB>(table + 16 * ((size_t)(e0 & 0x7FFFFFFFu) + i));
        for (u32 t = lo; t < hi; ++t) {
            const affine<FB> p = nxt;
            const u32 neg = e0 >> 31;
            const u32 e2 = list[min(t + 2, hi - 1)];
            nxt = aff_load<FB>(table + 16 * ((size_t)(e1 & 0x7FFFFFFFu) + i));
            e0 = e1;
            e1 = e2;
            if (!aff_is_identity(p)) {
                aff9<FB> q = aff9_unpack<FB>(p);
                if (neg) q.y = fe9_sub(fe9_zero(), q.y);
                xyzz9_madd<FB>(acc, q);
            }
        }
    }
    xyzz_store<FB>(sums + 32 * ((size_t)lb * nJ + i), xyzz9_is_identity(acc) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(acc));
}

// lane (i, v): sums[v * 8][i] <- sum_{b = 1..8} b * sums[v * 8 + b - 1][i]
template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse_windows(u32 *__restrict__ sums, u32 nJ) {
    H2_LATENCY_STAGE();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * nJ) return;
    const u32 v = t / nJ, i = t % nJ;
    xyzz<FB> run = xyzz_identity<FB>(), tot = xyzz_identity<FB>();
    for (int b = 8; b >= 1; --b) {
        xyzz_add<FB>(run, xyzz_load<FB>(sums + 32 * ((size_t)(v * 8 + b - 1) * nJ + i)));
        xyzz_add<FB>(tot, run);
    }
    xyzz_store<FB>(sums + 32 * ((size_t)(v * 8) * nJ + i), tot);
}

template <int FB>
__global__ void __launch_bounds__(256) ipa_collapse_finish(const u32 *__restrict__ sums, u32 nJ, u32 *__restrict__ out_xy) {
    H2_LATENCY_STAGE();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nJ) return;
    xyzz<FB> acc = xyzz_load<FB>(sums + 32 * ((size_t)24 * nJ + i));
    for (int v = 2; v >= 0; --v) {
        for (int d = 0; d < 4; ++d) acc = xyzz_dbl<FB>(acc);
        xyzz_add<FB>(acc, xyzz_load<FB>(sums + 32 * ((size_t)(v * 8) * nJ + i)));
    }
    const affine<FB> a = xyzz_to_affine<FB>(acc);
    fe_store(out_xy + 16 * (size_t)i, a.x);
    fe_store(out_xy + 16 * (size_t)i + 8, a.y);
}

// ---- the read-out with 8-bit sub-digits (the shipped form; H2_READOUT_NIBBLES=1 keeps the one above for A/B) ----------------
// |d_w| = e_0 + 256 e_1 with e_0 in [-127, 128], e_1 in [0, 128]: 2 x 128 lists instead of 4 x 8, and 2^J * 32 terms per output
// instead of 2^J * 64.  So that the 256 bucket sums per output neither travel through memory nor cost a lane each, a lane owns
// SIXTEEN consecutive magnitudes of one position (workgroup row y = position * 8 + g: magnitudes 16 g + 1 .. 16 g + 16), walks
// their lists from the largest down and keeps the two running sums of the bucket method in registers:
//     run += B_v;  tot += run      =>      tot = sum_r r * B_{16 g + r},   run = sum_r B_{16 g + r}
// (control flow and list reads are uniform across a wave, the gathers coalesced, as above).  ipa_readout_combine then forms, per
// output and position, sum_g tot_g + 16 * sum_g g * run_g (a second running sum over the 8 groups), ipa_readout_finish
// P_0 + 256 P_1 and the affine result.  Per output at J = 6: ~1800 mixed additions + 512 full ones against ~3800 + 64.
template <int FB>
__global__ void __launch_bounds__(256, 2) ipa_readout_groups(const u32 *__restrict__ table, const u32 *__restrict__ list,
                                                             const u32 *__restrict__ list_start, u32 nJ, u32 *__restrict__ sums) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (i >= nJ) return;
    const u32 first = (y >> 3) * 128 + (y & 7) * 16;          // the list of magnitude 16 g + 1 at this position
    xyzz9<FB> run = xyzz9_identity<FB>(), tot = xyzz9_identity<FB>();
    for (int r = 15; r >= 0; --r) {
        const u32 lo = list_start[first + r], hi = list_start[first + r + 1];
        xyzz9<FB> acc = xyzz9_identity<FB>();
        if (lo < hi) {
            u32 e0 = list[lo], e1 = list[min(lo + 1, hi - 1)];
            affine<FB> nxt = aff_load<FB>(table + 16 * ((size_t)(e0 & 0x7FFFFFFFu) + i));
            for (u32 t = lo; t < hi; ++t) {
                const affine<FB> p = nxt;
                const u32 neg = e0 >> 31;
                const u32 e2 = list[min(t + 2, hi - 1)];
                nxt = aff_load<FB>(table + 16 * ((size_t)(e1 & 0x7FFFFFFFu) + i));
                e0 = e1;
                e1 = e2;
                if (!aff_is_identity(p)) {
                    aff9<FB> q = aff9_unpack<FB>(p);
                    if (neg) q.y = fe9_sub(fe9_zero(), q.y);
                    xyzz9_madd<FB>(acc, q);
                }
            }
        }
        xyzz9_add<FB>(run, acc);
        xyzz9_add<FB>(tot, run);
    }
    xyzz_store<FB>(sums + 32 * ((size_t)(2 * y) * nJ + i), xyzz9_is_identity(tot) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(tot));
    xyzz_store<FB>(sums + 32 * ((size_t)(2 * y + 1) * nJ + i), xyzz9_is_identity(run) ? xyzz_identity<FB>() : xyzz9_to_r256<FB>(run));
}
// quad (i, position): sums[32 + position][i] <- sum_g tot_g + 16 * sum_g g * run_g.  27 dependent point operations per output and position,
// 2 nJ chains: the chip is nearly idle in them, so each runs on a QUAD of lanes (curve_wide.cuh; round 5: 0.22 -> ~0.08 ms at 2^14 outputs)
template <int FB>
__global__ void __launch_bounds__(256) ipa_readout_combine(u32 *__restrict__ sums, u32 nJ) {
    H2_LATENCY_STAGE();
    const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (t >= 2 * nJ) return;
    const u32 pos = t / nJ, i = t % nJ;
    xyzz<FB> P = xyzz_identity<FB>(), rr = xyzz_identity<FB>(), tt = xyzz_identity<FB>();
    for (int g = 7; g >= 0; --g) {
        const u32 y = pos * 8 + g;
        xyzz_add_wide<FB>(P, xyzz_load<FB>(sums + 32 * ((size_t)(2 * y) * nJ + i)));
        if (g) {
            xyzz_add_wide<FB>(rr, xyzz_load<FB>(sums + 32 * ((size_t)(2 * y + 1) * nJ + i)));
            xyzz_add_wide<FB>(tt, rr);
        }
    }
    for (int d = 0; d < 4; ++d) tt = xyzz_dbl_wide<FB>(tt);
    xyzz_add_wide<FB>(P, tt);
    if ((threadIdx.x & (kGroup - 1)) == 0) xyzz_store<FB>(sums + 32 * ((size_t)(32 + pos) * nJ + i), P);
}
template <int FB>
__global__ void __launch_bounds__(256) ipa_readout_finish(const u32 *__restrict__ sums, u32 nJ, u32 *__restrict__ out_xy) {
    H2_LATENCY_STAGE();
    const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    if (i >= nJ) return;
    xyzz<FB> acc = xyzz_load<FB>(sums + 32 * ((size_t)33 * nJ + i));
    for (int d = 0; d < 8; ++d) acc = xyzz_dbl_wide<FB>(acc);
    xyzz_add_wide<FB>(acc, xyzz_load<FB>(sums + 32 * ((size_t)32 * nJ + i)));
    if ((threadIdx.x & (kGroup - 1)) != 0) return;
    const affine<FB> a = xyzz_to_affine<FB>(acc);
    fe_store(out_xy + 16 * (size_t)i, a.x);
    fe_store(out_xy + 16 * (size_t)i + 8, a.y);
}


}  // namespace h2

using namespace h2;

extern "C" int h2_ipa_collapsed_generators_device(h2_bases_t basis, unsigned k, unsigned rounds, const uint64_t *challenges, int form,
                                                  void *d_out_xy, void *stream) {
    auto b = find_bases(basis);
    if (!b) return H2_ERR_HANDLE;
    if ((form != H2_FORM_CANONICAL && form != H2_FORM_MONTGOMERY) || !challenges || !d_out_xy || k < 1 || k > 26 || rounds < 1 ||
        rounds > k || rounds > 12 || b->n < ((size_t)1 << k) || b->c != 16 || b->W != 16)
        return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int sf = b->curve == H2_PALLAS ? H2_FQ : H2_FP;
    const u32 J = rounds, nJ = 1u << (k - J);
    u64 um[12 * 4];
    for (u32 r = 0; r < J; ++r) host_to_mont(sf, um + 4 * r, challenges + 4 * r, form);
    static const bool nibbles = [] { const char *e = ab_env("H2_READOUT_NIBBLES"); return e && e[0] == '1'; }();
    const int nlists = nibbles ? 32 : 256;
    std::vector<std::vector<u32>> lists(nlists);
    for (u32 h = 0; h < (1u << J); ++h) {
        u64 s[4], canon[4];
        memcpy(s, kHostField[sf].one, 32);
        for (u32 r = 0; r < J; ++r)
            if ((h >> (J - 1 - r)) & 1) host_mul(sf, s, s, um + 4 * r);          // the products of ipa_s_table
        host_from_mont(sf, canon, s);
        u32 carry = 0;
        for (u32 w = 0; w < 16; ++w) {
            u32 raw = (u32)((canon[w >> 2] >> (16 * (w & 3))) & 0xFFFFu) + carry;   // signed 16-bit digits, as msm_recode cuts them
            const bool neg = raw > 0x8000u;
            carry = neg ? 1 : 0;
            u32 mag = neg ? 0x10000u - raw : raw;                                 // |d| <= 2^15
            const u32 off = w * b->stride + h * nJ;
            if (!nibbles) {                                                       // |d| = e_0 + 256 e_1, e_0 in [-127, 128], e_1 in [0, 128]
                u32 e0 = mag & 255u, c8 = 0;
                bool e0neg = false;
                if (e0 > 128) {
                    e0 = 256 - e0;
                    e0neg = true;
                    c8 = 1;
                }
                const u32 e1 = (mag >> 8) + c8;                                   // <= 128: mag <= 2^15, and mag = 2^15 has e_0 = 0
                if (e0) lists[e0 - 1].push_back(off | ((neg != e0neg) ? 0x80000000u : 0u));
                if (e1) lists[128 + e1 - 1].push_back(off | (neg ? 0x80000000u : 0u));
                continue;
            }
            u32 c4 = 0;
            for (u32 v = 0; v < 4; ++v) {                                         // |d| = sum_v 16^v e_v, e_v in [-7, 8]
                u32 e = ((mag >> (4 * v)) & 15u) + c4;
                bool eneg = false;
                c4 = 0;
                if (e > 8) {
                    e = 16 - e;
                    eneg = true;
                    c4 = 1;
                }
                if (e) lists[v * 8 + e - 1].push_back(off | ((neg != eneg) ? 0x80000000u : 0u));
            }
            // c4 is 0 here: the top nibble of |d| <= 0x8000 is at most 8 with its carry
        }
        // carry is 0 here: the scalar is below 2^255, so the top digit takes it
    }
    std::vector<u32> flat, start(nlists + 1, 0);
    for (int l = 0; l < nlists; ++l) {
        start[l] = (u32)flat.size();
        flat.insert(flat.end(), lists[l].begin(), lists[l].end());
    }
    start[nlists] = (u32)flat.size();
    if (flat.empty()) flat.push_back(0);
    MsmContext &cx = msm_ctx(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.collapse.reserve((size_t)34 * nJ * 128)) != H2_OK) return rc;
    if ((rc = cx.collapse_list.reserve((nlists + 1) * 4 + flat.size() * 4)) != H2_OK) return rc;
    u32 *d_start = cx.collapse_list.as<u32>(), *d_list = d_start + nlists + 1;
    // pageable sources: consumed when hipMemcpyAsync returns; stream-ordered after the previous call's kernels
    H2_HIP(hipMemcpyAsync(d_start, start.data(), (nlists + 1) * 4, hipMemcpyHostToDevice, st));
    H2_HIP(hipMemcpyAsync(d_list, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, st));
    {   // the table's columns must be complete (a registration runs on the null stream and synchronises; nothing to wait for)
        dim3 blk(256), g1((nJ + 255) / 256, 32), g2((4 * nJ + 255) / 256), g3((nJ + 255) / 256);
        u32 *sums = cx.collapse.as<u32>();
        if (!nibbles) {
            dim3 r1((nJ + 255) / 256, 16), r2((2 * nJ * kGroup + 255) / 256);
            g3 = dim3((nJ * kGroup + 255) / 256);            // combine and finish run one chain per QUAD of lanes
            if (b->curve == H2_PALLAS) {
                hipLaunchKernelGGL((ipa_readout_groups<FP>), r1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
                hipLaunchKernelGGL((ipa_readout_combine<FP>), r2, blk, 0, st, sums, nJ);
                hipLaunchKernelGGL((ipa_readout_finish<FP>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
            } else {
                hipLaunchKernelGGL((ipa_readout_groups<FQ>), r1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
                hipLaunchKernelGGL((ipa_readout_combine<FQ>), r2, blk, 0, st, sums, nJ);
                hipLaunchKernelGGL((ipa_readout_finish<FQ>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
            }
        } else if (b->curve == H2_PALLAS) {
            hipLaunchKernelGGL((ipa_collapse_buckets<FP>), g1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
            hipLaunchKernelGGL((ipa_collapse_windows<FP>), g2, blk, 0, st, sums, nJ);
            hipLaunchKernelGGL((ipa_collapse_finish<FP>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
        } else {
            hipLaunchKernelGGL((ipa_collapse_buckets<FQ>), g1, blk, 0, st, (const u32 *)b->d_table, d_list, d_start, nJ, sums);
            hipLaunchKernelGGL((ipa_collapse_windows<FQ>), g2, blk, 0, st, sums, nJ);
            hipLaunchKernelGGL((ipa_collapse_finish<FQ>), g3, blk, 0, st, (const u32 *)sums, nJ, (u32 *)d_out_xy);
        }
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

