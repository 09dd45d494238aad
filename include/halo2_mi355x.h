/*
 * halo2_mi355x.h -- C ABI of libhalo2_mi355x.so: the MI355X (gfx950) implementation of the
 * halo2_proofs prover hot path (Pasta MSM + NTT).
 *
 * The reference (zcash/halo2, halo2_proofs 0.3.2) has no FFI seam of its own (`#![deny(unsafe_code)]`,
 * halo2_proofs/src/lib.rs:9).  The narrowest seam is the pair of free functions every MSM and FFT in
 * the crate funnels through -- `best_multiexp` (halo2_proofs/src/arithmetic.rs:143) and `best_fft`
 * (arithmetic.rs:192) -- plus the EvaluationDomain / Params wrappers directly above them.  Each entry
 * point below names the reference function whose body it replaces.  INTEGRATION.md shows the Rust
 * `extern "C"` block and shim a maintainer would add.
 *
 * Conventions
 *   field element : 4 x uint64_t little-endian limbs (32 bytes).
 *   affine point  : {x, y} = 8 x uint64_t (64 bytes); the identity is all-zero.
 *   Jacobian point: {X, Y, Z} = 12 x uint64_t (96 bytes); identity has Z = 0.  x = X/Z^2, y = Y/Z^3.
 *   `form`        : H2_FORM_MONTGOMERY -- limbs are in Montgomery form, R = 2^256: exactly the bytes a
 *                   Rust `Vec<Fp>` / `Vec<EpAffine>` holds in memory (zero-copy from pasta_curves);
 *                   H2_FORM_CANONICAL -- limbs are the canonical integer (`to_repr()` / `from_repr()`),
 *                   reachable from 100 % safe Rust.  Applies to inputs and outputs alike.
 *   curve id      : H2_PALLAS (coordinates in Fp, scalars in Fq) / H2_VESTA (coordinates in Fq, scalars Fp).
 *   field id      : H2_FP / H2_FQ.
 *   pointers      : `const uint64_t *` arguments named h_* / plain are HOST memory, borrowed for the call;
 *                   `d_*` arguments are DEVICE (HBM) pointers on the current HIP device.  The library never
 *                   frees or retains caller memory (registered bases are copied to the device).
 *   return value  : H2_OK or an H2_ERR_* code; never aborts.  The reference panics on bad lengths
 *                   (arithmetic.rs:144, :205); the Rust shim turns H2_ERR_ARGS back into a panic.
 *   threading     : all entry points are thread-safe and re-entrant (internal per-device lock).
 *   results       : bit-exact with the reference as group / field elements: canonical affine (x, y) of
 *                   an MSM and every canonical NTT output element are identical to the CPU path's.
 */
#ifndef HALO2_MI355X_H
#define HALO2_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2_OK 0
#define H2_ERR_ARGS 1   /* bad argument (null pointer, length mismatch, log_n out of range, bad id) */
#define H2_ERR_HIP 2    /* a HIP runtime call failed; see h2_last_error() */
#define H2_ERR_NODEV 3  /* no gfx950 device / HIP runtime unavailable */
#define H2_ERR_HANDLE 4 /* unknown or freed handle */
#define H2_ERR_DECODE 5 /* a compressed point does not decode (where pasta_curves' from_bytes returns None) */
#define H2_ERR_LOOKUP 6 /* permute_expression_pair: an input value does not occur in the table (Error::ConstraintSystemFailure) */
#define H2_ERR_PEER 7   /* a split multiexp / commit: ANOTHER rank failed its range; no result is written (a partial sum would be a wrong point) */

#define H2_FP 0
#define H2_FQ 1
#define H2_PALLAS 0
#define H2_VESTA 1
#define H2_FORM_CANONICAL 0
#define H2_FORM_MONTGOMERY 1
#define H2_OUT_JACOBIAN 0 /* 12 limbs, what `best_multiexp` returns (`C::Curve`) */
#define H2_OUT_AFFINE 1   /* 8 limbs, what `to_affine()` / `batch_normalize` would give */

typedef uint64_t h2_bases_t; /* opaque handle to a device-resident basis (Params::g / g_lagrange) */

/* ---- library ------------------------------------------------------------------------------ */
int h2_device_count(void);
/* The HIP device current on the calling thread (what registrations and launches will use), or -1 without a device. */
int h2_current_device(void);
/* Binds the calling thread to `device` (hipSetDevice) and warms the per-device context. */
int h2_init(int device);
/* Human-readable description of the last failure on this thread (static storage). */
const char *h2_last_error(void);
/* Returns the library's cached device scratch (per-stream multiexp / NTT workspaces, twiddle tables) of the current
 * device to the allocator after a device-wide synchronise.  Registered bases stay.  A long-lived prover that used many
 * streams calls this between phases; everything is re-created on demand. */
int h2_trim(void);
/* Window width the MSM would use for n points (informational; the result does not depend on it). */
int h2_msm_window_bits(size_t n);
/* The same for a registered basis of n points (h2_bases_register / h2_commit); h2_commit_pair_device: see h2_commit_pair_supported. */
int h2_commit_window_bits(size_t n);
/* 1 if h2_commit_pair_device accepts a registered basis of n points (n >= 8192 and the sort geometry fits), else 0. */
int h2_commit_pair_supported(size_t n);
/* Tuning knobs (never change results).  "msm_lane_fraction" in (0.05, 1]: share of the resident wave slots
 * one bucket-accumulation launch claims; < 1 lets commits issued on other streams overlap it (default 1).
 * "host_commit_chunk": scalars per range of h2_commit's pipelined transfer (0 = default 2^18 = 8 MiB). */
int h2_set_option(const char *key, double value);

/* ---- MSM: replaces best_multiexp (halo2_proofs/src/arithmetic.rs:143-180) -------------------- */
/* out = sum_i scalars[i] * bases[i].  n may be 0 (identity).  `out_kind` selects 12- or 8-limb output. */
int h2_msm(int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, int form,
           int out_kind, uint64_t *out);

/* Params::{new,read} keep `g` / `g_lagrange` for the life of the Params (poly/commitment.rs:26-33):
 * register them once, commit many times. */
int h2_bases_register(int curve, const uint64_t *bases_xy, size_t n, int form, h2_bases_t *handle);
/* the same from `n` affine points already in device memory (work that produces them must have completed) */
int h2_bases_register_device(int curve, const void *d_bases_xy, size_t n, int form, h2_bases_t *handle);
/* the same with the table's window width chosen by the caller (0 = the library's choice, as h2_bases_register; 4 .. 20, wider
 * than 16 only where the two-pass sort fits: H2_ERR_ARGS otherwise).  h2_commit_column_window_bits(n) is the width to ask for
 * when the table only serves independent column commits (Params::g / g_lagrange): 17 bits from 2^18 points on -- a scalar has
 * 15 digits instead of 16 -- which h2_commit_pair_device and h2_ipa_collapsed_generators_device do not take. */
int h2_bases_register_ex(int curve, const uint64_t *bases_xy, size_t n, int form, int window_bits, h2_bases_t *handle);
int h2_commit_column_window_bits(size_t n);
/* `Params::w` (poly/commitment.rs:26-33): installs the multiples of the affine point w_xy as the blind column of the handle's
 * table.  Compared by content with what the handle holds: the same point is a no-op, a different one waits for the device to
 * drain (commits with the old w may be in flight), installs it and returns when the column is complete.  Host pointer, blocking. */
int h2_bases_set_blind_base(h2_bases_t handle, const uint64_t *w_xy, int form);
/* What a handle holds: the number of registered points, the window width of its table, its curve (any pointer may be NULL). */
int h2_bases_info(h2_bases_t handle, size_t *n, int *window_bits, int *curve);
/* 1 if the handle has a blind base installed (h2_bases_set_blind_base, or a w_xy presented to a commit), 0 if not, a negative
 * status for a dead handle.  The rank-independent precondition of a blinded commit that is split over GPUs: every rank checks it
 * BEFORE entering the exchange step, so that either all ranks fail or none waits in a collective for one that never arrives. */
int h2_bases_blind_base_set(h2_bases_t handle);
int h2_bases_free(h2_bases_t handle);

/* replaces Params::commit / commit_lagrange (halo2_proofs/src/poly/commitment.rs:119-150):
 * out = sum_{i<n} scalars[i]*g[i] + blind*w.  The reference copies poly+blind and g+w into fresh
 * (n+1)-vectors per call; here g stays on the device and (w, blind) ride along.  `w_xy` / `blind`
 * may both be NULL for a plain MSM over the first n registered bases (IPA rounds,
 * poly/commitment/prover.rs:107-108); `blind` alone uses the handle's blind base (h2_bases_set_blind_base); a `w_xy` is compared
 * by content with it and installed first when it differs.  The column crosses PCIe in ranges that are committed as they land
 * (copy stream + two compute streams), so the transfer runs beside the bucket arithmetic. */
int h2_commit(h2_bases_t g, const uint64_t *scalars, size_t n, const uint64_t *w_xy,
              const uint64_t *blind, int form, int out_kind, uint64_t *out);

/* ---- NTT: replaces best_fft (halo2_proofs/src/arithmetic.rs:192-295) ------------------------- */
/* In place, natural order in and out: a[j] <- sum_i a[i] * omega^(i*j), n = 2^log_n, 0 <= log_n <= 32
 * (device memory permitting).  Reproduces the reference's butterfly network exactly, so the output
 * matches the CPU path for ANY omega, including the non-root omega of benches/fft.rs:17. */
int h2_ntt(int field, uint64_t *a, unsigned log_n, const uint64_t *omega, int form);

/* replaces EvaluationDomain::ifft (halo2_proofs/src/poly/domain.rs:375-383; lagrange_to_coeff :227):
 * best_fft with omega_inv, then every element times `divisor` (fused into the last pass). */
int h2_ifft(int field, uint64_t *a, unsigned log_n, const uint64_t *omega_inv, const uint64_t *divisor,
            int form);

/* replaces EvaluationDomain::coeff_to_extended (poly/domain.rs:241-255, :357-373):
 * out[i] = a[i] * {1, zeta, zeta^2}[i % 3] for i < 2^k, zero-extended to 2^ext_k, forward NTT with
 * extended_omega.  `a` has 2^k elements, `out` 2^ext_k (may not alias). */
int h2_coeff_to_extended(int field, const uint64_t *a, uint64_t *out, unsigned k, unsigned ext_k,
                         const uint64_t *g_coset, const uint64_t *g_coset_inv,
                         const uint64_t *extended_omega, int form);

/* replaces EvaluationDomain::extended_to_coeff (poly/domain.rs:303-325): inverse NTT of size 2^ext_k,
 * times extended_ifft_divisor, then a[i] *= {1, zeta^2, zeta}[i % 3].  In place; the caller truncates
 * to n*(degree-1) as the reference does (:321-322). */
int h2_extended_to_coeff(int field, uint64_t *a, unsigned ext_k, const uint64_t *g_coset,
                         const uint64_t *g_coset_inv, const uint64_t *extended_omega_inv,
                         const uint64_t *extended_ifft_divisor, int form);

/* replaces EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:329-348): a[i] *= t_evaluations[i mod nt]
 * over the extended domain (nt = 2^(ext_k - k) <= 4096 inverse vanishing-polynomial values, host memory). */
int h2_divide_by_vanishing_poly(int field, uint64_t *a, unsigned ext_k, const uint64_t *t_evaluations,
                                size_t nt, int form);

/* ---- device-resident variants (data already in HBM; `stream` is a hipStream_t or NULL) -------- */
/* Same contracts as above with device pointers; asynchronous on `stream`; outputs land in device
 * memory.  Used by batched provers and by bench.py (inputs resident in HBM before timing starts). */
int h2_msm_device(int curve, const void *d_scalars, const void *d_bases_xy, size_t n, int form,
                  int out_kind, void *d_out, void *stream);
/* Blind base: `Params::w` is a field of `Params`, fixed for its life (poly/commitment.rs:26-33, :102-103), and here a property of
 * the HANDLE: h2_bases_set_blind_base installs its multiples as one more column of the registered table.  A commit that passes
 * `d_blind` and NO `d_w_xy` uses that column (H2_ERR_ARGS if none was ever installed).  A commit may still present a `d_w_xy`:
 * its 64 BYTES -- never its address -- are compared with the installed point by one small kernel on `stream`, and only a
 * different point rebuilds the column (and becomes the handle's blind base); commits that use the old w on other streams
 * must have completed by then.  d_w_xy without d_blind: H2_ERR_ARGS.  Both NULL: no blind term. */
int h2_commit_device(h2_bases_t g, const void *d_scalars, size_t n, const void *d_w_xy,
                     const void *d_blind, int form, int out_kind, void *d_out, void *stream);
/* The commit restricted to registered bases [first, first + n): d_scalars[i] multiplies base first + i; d_blind (optional) adds
 * blind * (the handle's blind base).  One range of a commit split over GPUs or over the chunks of a host transfer. */
int h2_commit_range_device(h2_bases_t g, const void *d_scalars, size_t first, size_t n, const void *d_blind, int form,
                           int out_kind, void *d_out, void *stream);
/* TWO commits from one column over a registered basis of n points: column i < n - 4 feeds output (i >> pair_shift) & 1, the last
 * four columns feed outputs 0, 1, 0, 1.  This is one round of the opening argument written over the original generators
 * (poly/commitment/prover.rs:107-114): L_j and R_j have disjoint supports in g, so they share the scalar column produced by
 * h2_ipa_round_scalars_device (pass the same buffer for d_cl and d_cr), and the basis g || u || u || w || w carries the
 * [value z] U and [rand] W terms of each.  d_out: output 0 then output 1 (2 x 12 or 2 x 8 limbs).  n >= 8192. */
int h2_commit_pair_device(h2_bases_t g, const void *d_scalars, size_t n, unsigned pair_shift, int form, int out_kind,
                          void *d_out, void *stream);
/* `count` independent commits over one registered basis -- the column commits of a prover phase
 * (plonk/prover.rs:93-101, 301-313; vanishing/prover.rs:96-108).  d_scalars[i] / d_blinds[i] / d_outs[i] are
 * device pointers held in HOST arrays.  The library picks the faster of two forms by size and count: the COLUMN-BATCHED form --
 * one sort / accumulate / fold launch set for up to eight columns at a time, the column a grid dimension; a single group runs on
 * `stream` itself, several alternate over two internal streams -- or one commit per column spread over three internal streams
 * (many full-size columns: one column's sort / fold beside another's accumulate); either way the work is joined on `stream`.
 * d_blinds NULL: no blind term; d_blinds without d_w_xy: the handle's blind base; a d_w_xy is content-checked once, on
 * `stream`, as for h2_commit_device. */
int h2_commit_batch_device(h2_bases_t g, const void *const *d_scalars, size_t count, size_t n,
                           const void *d_w_xy, const void *const *d_blinds, int form, int out_kind,
                           void *const *d_outs, void *stream);
/* `count` independent multiexps over caller-supplied bases (e.g. the L_j / R_j pair of one opening-argument round,
 * halo2_proofs/src/poly/commitment/prover.rs:107-108), overlapped on internal streams and joined on `stream`.
 * Arrays of `count` device pointers / lengths; argument meaning per entry as h2_msm_device. */
int h2_msm_batch_device(int curve, const void *const *d_scalars, const void *const *d_bases_xy, const size_t *n,
                        size_t count, int form, int out_kind, void *const *d_outs, void *stream);
int h2_ntt_device(int field, void *d_a, unsigned log_n, const uint64_t *omega, int form, void *stream);
int h2_ifft_device(int field, void *d_a, unsigned log_n, const uint64_t *omega_inv,
                   const uint64_t *divisor, int form, void *stream);
/* `count` independent transforms of one size (the column FFTs of a prover phase, halo2_proofs/src/plonk/prover.rs:111-117,
 * 322-327) in one call: overlapped on internal streams, joined on `stream`.  d_a: array of `count` device pointers; each
 * vector is transformed in place as by h2_ntt_device / h2_ifft_device. */
int h2_ntt_batch_device(int field, void *const *d_a, size_t count, unsigned log_n, const uint64_t *omega, int form,
                        void *stream);
int h2_ifft_batch_device(int field, void *const *d_a, size_t count, unsigned log_n, const uint64_t *omega_inv,
                         const uint64_t *divisor, int form, void *stream);
int h2_coeff_to_extended_device(int field, const void *d_a, void *d_out, unsigned k, unsigned ext_k,
                                const uint64_t *g_coset, const uint64_t *g_coset_inv,
                                const uint64_t *extended_omega, int form, void *stream);
int h2_extended_to_coeff_device(int field, void *d_a, unsigned ext_k, const uint64_t *g_coset,
                                const uint64_t *g_coset_inv, const uint64_t *extended_omega_inv,
                                const uint64_t *extended_ifft_divisor, int form, void *stream);

int h2_divide_by_vanishing_poly_device(int field, void *d_a, unsigned ext_k, const uint64_t *t_evaluations,
                                       size_t nt, int form, void *stream);

/* Sum of `count` Jacobian points laid out contiguously (12 limbs each, Montgomery) -> one Jacobian
 * point.  The local step after the 96-byte all-gather of a range-split MSM (one partial per GPU). */
int h2_points_sum(int curve, const uint64_t *points_xyz, size_t count, uint64_t *out_xyz);
/* The same over device memory, asynchronous on `stream`; `form` applies to inputs and output, `out_kind` as for h2_msm. */
int h2_points_sum_device(int curve, const void *d_points_xyz, size_t count, int form, int out_kind, void *d_out, void *stream);

/* ---- several GPUs (no counterpart in the reference, whose create_proof is one process) ----------- */
/* The `count` independent column commits of a prover phase (plonk/prover.rs:93-101, 301-313; vanishing/prover.rs:96-108)
 * spread over `ndev` devices from ONE process: column i goes to devices[i % ndev], which holds handles[i % ndev] -- the same
 * bases registered once per device (h2_init(dev) + h2_bases_register on each).  Host columns in, host points out; one host
 * thread and three streams per device, no collective (a result is one point).  Blocking. */
int h2_commit_batch_multi(const h2_bases_t *handles, const int *devices, int ndev, const uint64_t *const *scalars,
                          size_t count, size_t n, const uint64_t *w_xy, const uint64_t *const *blinds, int form,
                          int out_kind, uint64_t *const *outs);
/* ONE multiexp (best_multiexp, arithmetic.rs:143) cut into ndev contiguous point ranges, one per device; the ndev partial
 * points (96 bytes each) are brought to devices[0] and added there.  Host pointers, blocking. */
int h2_msm_split_multi(int curve, const uint64_t *scalars, const uint64_t *bases_xy, size_t n, const int *devices,
                       int ndev, int form, int out_kind, uint64_t *out);
/* The same split for the one-process-per-GPU model: RCCL (bound with dlopen at first use) carries the exchange step.
 * Rank 0 calls h2_rccl_unique_id and hands the 128 bytes to the other ranks by whatever means its launcher offers
 * (a broadcast of its process group, MPI, a file); every rank then calls h2_rccl_init(id, rank, world) with its GPU current.
 * h2_msm_split_rccl_device: every rank passes the same device-resident problem; rank r multiplies points
 * [n r / world, n (r + 1) / world), ONE ncclAllGather of a 128-byte slot per rank (the 96-byte partial + a status word), and
 * every rank writes the total to d_out.  A rank whose range fails still enters the exchange with its status set: it returns its
 * own error and EVERY other rank returns H2_ERR_PEER -- nobody hangs, nobody sums the partials that did arrive.  The status read-back
 * synchronises `stream` once per call (the sum itself is enqueued behind it). */
int h2_rccl_unique_id(uint8_t id_out[128]);
int h2_rccl_init(const uint8_t id[128], int rank, int world);
int h2_rccl_finalize(void);
int h2_msm_split_rccl_device(int curve, const void *d_scalars, const void *d_bases_xy, size_t n, int form,
                             int out_kind, void *d_out, void *stream);
/* The same exchange for a commit over REGISTERED bases (Params::commit, an opening-argument round; BASELINE configs[4]'s
 * "RCCL-summed final commit"): every rank holds the table and the column, rank r commits table columns
 * [n r / world, n (r + 1) / world), the last rank carries the blind term (the handle's blind base), ONE ncclAllGather (128-byte
 * slots: partial + status, as above -- H2_ERR_PEER on every rank when any rank failed), every rank writes the total to d_out. */
int h2_commit_split_rccl_device(h2_bases_t g, const void *d_scalars, size_t n, const void *d_blind, int form, int out_kind,
                                void *d_out, void *stream);

/* ---- IPA round kernels (next to the MSMs inside commitment::create_proof) ----------------------- */
/* replaces parallel_generator_collapse (halo2_proofs/src/poly/commitment/prover.rs:154-166):
 * g holds 2*half affine points; on return g[i] = g[i] + [challenge] * g[half + i] for i < half, affine
 * (the caller truncates to `half`, :137).  `challenge` is a scalar-field element in `form`. */
int h2_generator_collapse(int curve, uint64_t *g_xy, size_t half, const uint64_t *challenge, int form);
int h2_generator_collapse_device(int curve, void *d_g_xy, size_t half, const uint64_t *challenge, int form,
                                 void *stream);
/* replaces the p' / b collapse loop (prover.rs:128-131): a[i] += a[half + i] * factor for i < half. */
int h2_fold_scalars(int field, uint64_t *a, size_t half, const uint64_t *factor, int form);
int h2_fold_scalars_device(int field, void *d_a, size_t half, const uint64_t *factor, int form, void *stream);
/* The scalars of round j's L_j / R_j multiexps (prover.rs:107-108) over the ORIGINAL generators instead of the collapsed
 * G': after j collapses G'[i] = sum_h s_j(h) * G[i + h * 2^(k-j)], where s_j(h) is the product of the challenges u_r, r < j,
 * selected by the bits of h (bit j-1-r <-> u_r; the products compute_s builds for the verifier, verifier.rs:156-172).  So with
 * m = h * 2^(k-j) + i and half = 2^(k-j-1):
 *     L_j = <p'_hi, G'_lo> = sum_m cl[m] * G[m],   cl[m] = p'[half + i] * s_j(h) for i <  half, else 0
 *     R_j = <p'_lo, G'_hi> = sum_m cr[m] * G[m],   cr[m] = p'[i - half] * s_j(h) for i >= half, else 0
 * Both become commits over a registered G (h2_commit_batch_device): every round costs two half-empty registered multiexps,
 * and the generator collapse (prover.rs:136-137, 2^(k-j-1) scalar multiplications per round) is never needed.
 * d_p: the current p' (2^(k-j) elements); challenges: u_0 .. u_{j-1} (host, `form`; may be NULL when j = 0);
 * d_cl, d_cr: 2^k elements each, in the form of d_p.  1 <= k <= 30, j < k. */
int h2_ipa_round_scalars_device(int field, const void *d_p, unsigned k, unsigned j, const uint64_t *challenges, int form,
                                void *d_cl, void *d_cr, void *stream);
/* The whole round loop of commitment::create_proof (prover.rs:104-142) in one call, p' and b resident: per round the two inner
 * products (:109-110), the L_j / R_j scalars over the original generators (above), their commit, the two points to the
 * transcript (:121-122), the challenge and its inverse (:124-125), the p' / b folds (:128-133) and the blind bookkeeping
 * (:140-141).  The host is visited once per round -- L_j and R_j (192 bytes) land in pinned memory, are normalised with one
 * shared inversion and handed to the caller's transcript through the two TranscriptWrite methods the loop uses:
 *     write_point(user, xy)  the affine point, 8 x u64 Montgomery; returns H2_OK or an error the call passes on
 *     squeeze(user, out)     the next challenge scalar, 4 x u64 Montgomery, into out
 * Everything here is Montgomery form (the working form of resident vectors).
 *   paired != 0: `basis` is a registered g || u || u || w || w (2^k + 4 points; h2_commit_pair_supported) and each
 *                round is ONE h2_commit_pair_device over d_column_l (2^k + 4 scalars of scratch); d_column_r unused
 *   paired == 0: `basis` is a registered g || u || w (2^k + 2 points), each round two commits (h2_commit_batch_device) over
 *                d_column_l / d_column_r (2^k + 2 scalars of scratch each): any table size
 * d_p: p' (2^k scalars, folded in place; c_out receives the final p'[0]);  d_b: b, likewise;  z: the challenge of :66;
 * rands: l_0, r_0, l_1, r_1, ... (2k scalars, the order the reference draws them, :111-112);
 * f_out: sum_j (l_j u_j^-1 + r_j u_j), the amount the synthetic blinding factor grows by (:140-141).
 * switch_rounds = J > 0 (paired only, J < k, J <= 12, 16-bit window table): after J rounds the argument leaves the original
 *   generators -- G'_J is read off the table (h2_ipa_collapsed_generators_device), registered next to u and w (uw_xy: u then w,
 *   host, affine Montgomery) as a table of 2^(k-J) points, and the remaining rounds run over it: a round over the original
 *   generators costs a full-size commit whatever j, a round over G'_J a small one.  0 = never; H2_IPA_SWITCH_DEFAULT = the
 *   library's choice (h2_ipa_default_switch_rounds: from k = 16 on the rounds that leave a table of 2^14 points up to k = 20, six rounds at
 *   k = 21, five beyond).  Same L_j, R_j, so the same proof bytes.
 * Returns H2_ERR_ARGS if an L_j / R_j is the point at infinity or a challenge is zero (the reference errors / panics).
 * h2_ipa_rounds: the same with p' and b in host memory (copied in; only c and f leave the argument). */
#define H2_IPA_SWITCH_DEFAULT 0xFFFFFFFFu
typedef int (*h2_ipa_write_point_fn)(void *user, const uint64_t *xy);
typedef int (*h2_ipa_squeeze_fn)(void *user, uint64_t *challenge);
unsigned h2_ipa_default_switch_rounds(unsigned k, int paired);
int h2_ipa_rounds_device(int curve, unsigned k, unsigned switch_rounds, h2_bases_t basis, int paired, void *d_p, void *d_b,
                         const uint64_t *z, const uint64_t *rands, const uint64_t *uw_xy, void *d_column_l, void *d_column_r,
                         h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze, void *user, uint64_t *c_out,
                         uint64_t *f_out, void *stream);
int h2_ipa_rounds(int curve, unsigned k, unsigned switch_rounds, h2_bases_t basis, int paired, const uint64_t *p, const uint64_t *b,
                  const uint64_t *z, const uint64_t *rands, const uint64_t *uw_xy, h2_ipa_write_point_fn write_point,
                  h2_ipa_squeeze_fn squeeze, void *user, uint64_t *c_out, uint64_t *f_out);
/* replaces poly::commitment::prover::create_proof (halo2_proofs/src/poly/commitment/prover.rs:26-151) -- the whole opening
 * argument for p_poly at x_3 -- as ONE call: the function a Rust shim swaps, with the caller keeping what is the caller's (its rng
 * and its transcript).  Everything is Montgomery form.
 *   g_basis        Params::g registered with Params::w installed (h2_bases_set_blind_base): the commitment to s_poly (:56)
 *   opening_basis  g || u || u || w || w (paired != 0) or g || u || w (paired == 0), as for h2_ipa_rounds_device; switch_rounds,
 *                  uw_xy: as there
 *   p_poly         the 2^k coefficients (:41), read only;  p_blind (:38), x3 (:39): one scalar each (host)
 *   s_poly         2^k fresh random scalars, s_blind one more, rands the 2k blinds l_0, r_0, l_1, ... -- C::Scalar::random in the
 *                  order the reference draws them (:45-47, :53, :111-112).  h2_open_device overwrites d_s_poly (it becomes p' and is
 *                  folded in place); h2_open reads s_poly and p_poly from host memory, once each, and writes nothing back
 *   write_point    receives the commitment to s_poly (:57), then L_j, R_j per round (:121-122); squeeze yields xi (:62), z (:66),
 *                  then u_j per round (:124) -- the TranscriptWrite calls of the function, in its order
 *   c_out, f_out   the two scalars the caller writes last (:146-148): the final p'[0] and the synthetic blinding factor
 * The constant-coefficient corrections (:51, :72) happen on the device; b (:86-97) and v run beside the commitment to s_poly
 * (v = p_poly(x_3): s_poly(x_3) is exactly zero after :51).  Scratch (b, the round column, landing places for host vectors)
 * belongs to the (device, stream) context and goes back with h2_trim.  Same proof bytes as the reference for the same
 * randomness.  Errors: as h2_ipa_rounds_device; H2_ERR_ARGS also when g_basis has no blind base or a table has the wrong size --
 * all of that before anything reaches the transcript. */
int h2_open_device(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                   const uint64_t *uw_xy, const void *d_p_poly, const uint64_t *p_blind, const uint64_t *x3, void *d_s_poly,
                   const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze,
                   void *user, uint64_t *c_out, uint64_t *f_out, void *stream);
/* p_poly resident in HBM, the fresh s_poly where a host rng leaves it (what a prover that keeps its polynomials on the device has at this point):
 * s_poly crosses PCIe inside the call, in quarters, each committed as it lands (from k = 16 on). */
int h2_open_device_host_s(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
                          const uint64_t *uw_xy, const void *d_p_poly, const uint64_t *p_blind, const uint64_t *x3, const uint64_t *s_poly,
                          const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze,
                          void *user, uint64_t *c_out, uint64_t *f_out, void *stream);
int h2_open(int curve, unsigned k, h2_bases_t g_basis, h2_bases_t opening_basis, int paired, unsigned switch_rounds,
            const uint64_t *uw_xy, const uint64_t *p_poly, const uint64_t *p_blind, const uint64_t *x3, const uint64_t *s_poly,
            const uint64_t *s_blind, const uint64_t *rands, h2_ipa_write_point_fn write_point, h2_ipa_squeeze_fn squeeze, void *user,
            uint64_t *c_out, uint64_t *f_out);
/* The generators after `rounds` collapses, without collapsing: G'[i] = sum_{h < 2^rounds} s(h) * G[i + h * 2^(k-rounds)] for
 * i < 2^(k-rounds), s(h) the challenge products of h2_ipa_round_scalars_device, read off the registered table of `basis` (its
 * first 2^k points are G; the table must use 16-bit windows, h2_commit_window_bits) as 2^(k-rounds) multiexps that share their
 * scalars: every 16-bit table digit as two signed 8-bit sub-digits, ~28 * 2^k mixed additions plus 512 full additions per output
 * for the bucket weights, no sort (csrc/msm.hip, ipa_readout_*).  With h2_bases_register_device this is how
 * h2_ipa_rounds_device moves to a 2^rounds times smaller table after its first rounds.
 * challenges: u_0 .. u_{rounds-1} in `form` (host); d_out_xy: 2^(k-rounds) affine points, Montgomery form.  rounds <= 12. */
int h2_ipa_collapsed_generators_device(h2_bases_t basis, unsigned k, unsigned rounds, const uint64_t *challenges, int form,
                                       void *d_out_xy, void *stream);

/* ---- the Fiat-Shamir transcript (host side) ------------------------------------------------------- */
/* replaces Blake2bWrite<_, C, Challenge255<C>> (halo2_proofs/src/transcript.rs:150-198, 286-296) for callers that want the
 * round loop above to stay in native code: BLAKE2b-512 personalised "Halo2-Transcript"; points are absorbed as canonical
 * (x, y) and written compressed (x with the sign of y in the top bit), scalars canonical little-endian; a challenge is the
 * digest of a copy of the state reduced from 512 bits into the curve's scalar field.  Inputs and challenges are Montgomery
 * limbs; `jacobian` != 0: the point is (X, Y, Z), 12 limbs, and is normalised first (the prover's .to_affine()).  Writing the
 * point at infinity returns H2_ERR_ARGS (the reference returns an io::Error, :209-214).  Host arithmetic only. */
typedef uint64_t h2_transcript_t;
int h2_transcript_new(int curve, h2_transcript_t *t);
int h2_transcript_free(h2_transcript_t t);
int h2_transcript_common_point(h2_transcript_t t, const uint64_t *point, int jacobian);
int h2_transcript_write_point(h2_transcript_t t, const uint64_t *point, int jacobian);
int h2_transcript_common_scalar(h2_transcript_t t, const uint64_t *scalar);
int h2_transcript_write_scalar(h2_transcript_t t, const uint64_t *scalar);
int h2_transcript_squeeze_challenge(h2_transcript_t t, uint64_t *challenge);
/* the bytes written so far (the proof): *len always receives their count, out is filled when cap >= *len */
int h2_transcript_bytes(h2_transcript_t t, uint8_t *out, size_t cap, size_t *len);
/* write_point / squeeze of h2_ipa_rounds_device over such a transcript: pass these with user = (void *)(uintptr_t)t */
int h2_transcript_cb_write_point(void *user, const uint64_t *xy);
int h2_transcript_cb_squeeze(void *user, uint64_t *challenge);

/* ---- Params set-up: Lagrange basis by an FFT over curve points -------------------------------- */
/* replaces the point FFT + 2^-k scaling + batch_normalize of Params::new
 * (halo2_proofs/src/poly/commitment.rs:77-100): out[j] = 2^-k * sum_i alpha_inv^(i*j) * g[i], affine, where
 * alpha_inv is the inverse 2^k-th root of unity of the curve's scalar field.  g and out hold 2^k points. */
int h2_lagrange_basis(int curve, const uint64_t *g_xy, uint64_t *out_xy, unsigned k, int form);
int h2_lagrange_basis_device(int curve, const void *d_g_xy, void *d_out_xy, unsigned k, int form, void *stream);

/* ---- polynomial helpers between the commits and the transforms ------------------------------- */
/* Each replaces one O(n) field loop of the prover so a column can stay in HBM from witness to opening.
 * Vectors are n contiguous 32-byte elements in `form`; single field elements (`point`, `x`, `init`) are
 * host pointers in `form`; the *_device variants take device pointers for vectors and results. */
/* eval_polynomial (halo2_proofs/src/arithmetic.rs:298-303): out = sum_i poly[i] * point^i. */
int h2_eval_polynomial(int field, const uint64_t *poly, size_t n, const uint64_t *point, int form, uint64_t *out);
int h2_eval_polynomial_device(int field, const void *d_poly, size_t n, const uint64_t *point, int form, void *d_out,
                              void *stream);
/* compute_inner_product (arithmetic.rs:308-318): out = sum_i a[i] * b[i]; both vectors have n elements
 * (the reference panics on a length mismatch; the caller passes one n). */
int h2_inner_product(int field, const uint64_t *a, const uint64_t *b, size_t n, int form, uint64_t *out);
int h2_inner_product_device(int field, const void *d_a, const void *d_b, size_t n, int form, void *d_out, void *stream);
/* kate_division (arithmetic.rs:322-341): out[0 .. n-2] = quotient of a(X) by (X - point), remainder dropped.
 * n >= 1 (the reference underflows on an empty input); out must not alias a. */
int h2_kate_division(int field, const uint64_t *a, size_t n, const uint64_t *point, int form, uint64_t *out);
int h2_kate_division_device(int field, const void *d_a, size_t n, const uint64_t *point, int form, void *d_out, void *stream);
/* the `b` vector of the opening argument (poly/commitment/prover.rs:90-97): out[i] = x^i, i < n. */
int h2_powers(int field, const uint64_t *x, size_t n, int form, uint64_t *out);
int h2_powers_device(int field, const uint64_t *x, size_t n, int form, void *d_out, void *stream);
/* `s_poly * xi + p_poly` (poly/commitment/prover.rs:70): a[i] = a[i] * x + b[i]. */
int h2_scale_add(int field, uint64_t *a, const uint64_t *x, const uint64_t *b, size_t n, int form);
int h2_scale_add_device(int field, void *d_a, const uint64_t *x, const void *d_b, size_t n, int form, void *stream);
/* ff::BatchInvert as the grand products use it (plonk/permutation/prover.rs:118, plonk/lookup/prover.rs:297):
 * a[i] = 1 / a[i]; zeros are left zero. */
int h2_batch_invert(int field, uint64_t *a, size_t n, int form);
int h2_batch_invert_device(int field, void *d_a, size_t n, int form, void *stream);
/* the running product of a permutation / lookup argument (plonk/permutation/prover.rs:147-153,
 * plonk/lookup/prover.rs:318-326): z[0] = init, z[i] = z[i-1] * m[i-1] for 0 < i < n.  m holds at least
 * n - 1 factors; z has n elements and must not alias m. */
int h2_grand_product(int field, const uint64_t *m, size_t n, const uint64_t *init, int form, uint64_t *z);
int h2_grand_product_device(int field, const void *d_m, size_t n, const uint64_t *init, int form, void *d_z, void *stream);

/* ---- expression evaluation between the coset FFTs and the quotient iFFT ----------------------- */
/* poly::Evaluator::evaluate (halo2_proofs/src/poly/evaluator.rs:129-228): one expression tree over registered
 * polynomials, evaluated element by element in a single kernel.  The tree arrives flattened in post-order as 32-bit
 * words `op | operand << 8`:
 *   H2_EV_POLY   operand = polynomial index; the NEXT word is the signed element shift (rotation * 1 for the Lagrange
 *                basis, * 2^(extended_k - k) for the extended basis; must be 0 in the coefficient basis, evaluator.rs:519)
 *   H2_EV_CONST  operand = constant index (Ast::ConstantTerm)      H2_EV_LINEAR  operand = constant index (Ast::LinearTerm:
 *                value consts[c] * omega^i; the caller folds ZETA into the constant for the extended basis, :590-600)
 *   H2_EV_ADD, H2_EV_MUL (element-wise; Lagrange and extended bases, as in the reference: evaluator.rs:370-418), H2_EV_SCALE operand = constant index,
 *   H2_EV_MULADD operand = constant index of the base: one fold step acc * base + term of Ast::DistributePowers (:186-196)
 * basis: 0 coefficient, 1 Lagrange, 2 extended Lagrange.  consts: n_consts field elements, d_polys: n_polys device
 * vectors of 2^log_len elements, all in MONTGOMERY form (products of canonical-form data would not be canonical).
 * omega: the domain's (extended) root of unity, needed only when the program has a LINEAR node.  Stack depth <= 9. */
#define H2_EV_POLY 1
#define H2_EV_CONST 2
#define H2_EV_LINEAR 3
#define H2_EV_ADD 4
#define H2_EV_MUL 5
#define H2_EV_SCALE 6
#define H2_EV_MULADD 7
int h2_evaluate_device(int field, int basis, const uint32_t *program, size_t n_words, const uint64_t *consts, size_t n_consts,
                       const void *const *d_polys, size_t n_polys, unsigned log_len, const uint64_t *omega, void *d_out,
                       void *stream);

/* ---- lookup argument: the data-dependent step (plonk/lookup/prover.rs:557-647) ----------------------------------- */
/* `Vec<F>::sort()` as the prover uses it (:574): ascending by canonical value, in place.  n need not be a power of two. */
int h2_sort_device(int field, void *d_a, size_t n, int form, void *stream);
/* replaces permute_expression_pair over the `n` usable rows (the caller appends the blinding rows, :624-627):
 * permuted_input = the input values sorted ascending; permuted_table[i] = permuted_input[i] on the first row of every run
 * of equal input values, and the table values not consumed that way -- ascending -- on the repeated rows taken from the
 * last one up (the reference pops them off the end of its list, :617-622).  Returns H2_ERR_LOOKUP when an input value does
 * not occur in the table (Error::ConstraintSystemFailure, :609-611).  Synchronises `stream` (the status has to reach the
 * host).  Inputs are not modified; outputs hold n elements each and may not alias the inputs. */
int h2_permute_expression_pair_device(int field, const void *d_input, const void *d_table, size_t n, int form,
                                      void *d_permuted_input, void *d_permuted_table, void *stream);

/* ---- compressed points: the URS file and proof encoding --------------------------------------- */
/* pasta_curves `to_bytes` as Params::write uses it (halo2_proofs/src/poly/commitment.rs:169-181) and write_point
 * (transcript.rs:183-187): out[32 i ..] = x little-endian with the parity of y in bit 255; identity = 32 zero bytes.
 * Input: n affine points in `form`. */
int h2_points_compress(int curve, const uint64_t *xy, size_t n, int form, uint8_t *out_bytes);
int h2_points_compress_device(int curve, const void *d_xy, size_t n, int form, void *d_out_bytes, void *stream);
/* `from_bytes` as Params::read uses it (commitment.rs:184-205): one base-field square root per point.  Returns
 * H2_ERR_DECODE if any encoding is invalid (x >= p, x^3 + 5 not a square, or (0, odd)); those entries are zeroed.
 * The device variant synchronises `stream` to learn the verdict. */
int h2_points_decompress(int curve, const uint8_t *bytes, size_t n, int form, uint64_t *out_xy);
int h2_points_decompress_device(int curve, const void *d_bytes, size_t n, int form, void *d_out_xy, void *stream);

/* ---- hash-to-curve: `C::CurveExt::hash_to_curve(domain_prefix)` as Params::new calls it (poly/commitment.rs:52-62, :102-104) ---- */
/* out[i] = hash_to_curve(domain_prefix)(message i), affine; `count` messages of `msg_len` (<= 64) bytes each, contiguous.  The map
 * is pasta_curves' (BLAKE2b-512 XMD, simplified SWU on iso-Pallas / iso-Vesta, 3-isogeny: Zcash protocol specification 5.4.9.8);
 * Params::new(k) is count = 2^k messages {0, i as LE u32} plus {1} for w and {2} for u with the prefix "Halo2-Parameters". */
int h2_hash_to_curve(int curve, const char *domain_prefix, const uint8_t *msgs, size_t msg_len, size_t count, int form,
                     uint64_t *out_xy);
int h2_hash_to_curve_device(int curve, const char *domain_prefix, const void *d_msgs, size_t msg_len, size_t count,
                            int form, void *d_out_xy, void *stream);

/* ---- measurement aid (no reference counterpart) ------------------------------------------------ */
/* When enabled, the library brackets its dominant kernels with HIP events on the launching stream.
 * h2_profile_read drains them: slot 0 = MSM bucket accumulation, 1 = NTT passes (sum over the passes
 * of one transform counts as several launches), 2 = MSM recode+sort, 3 = MSM reduce+combine. */
#define H2_PROF_MSM_ACCUMULATE 0
#define H2_PROF_NTT_PASS 1
#define H2_PROF_MSM_SORT 2
#define H2_PROF_MSM_REDUCE 3
/* on = 1: every slot records; on = 2: only the dominant kernels (H2_PROF_MSM_ACCUMULATE, H2_PROF_NTT_PASS) -- an event pair costs the
 * launching stream ~10 us, and bench.py's timed region wants the accumulate's duration without paying for the sort's and the fold's;
 * on = 0: off. */
int h2_profile_enable(int on);
int h2_profile_read(int slot, double *total_ms, uint64_t *launches);
/* The same, plus `busy_ms`: the length of the union of the launch intervals since h2_profile_enable(1).  With launches
 * issued on several streams overlapping on the device, total_ms counts shared time once per launch; busy_ms is the time
 * the device spent on the kernel, so busy_ms / launches never exceeds the wall time per launch. */
int h2_profile_read_busy(int slot, double *total_ms, double *busy_ms, uint64_t *launches);
/* ENVIRONMENT.  libhalo2_mi355x.so reads ONE environment variable, the diagnostic H2_TIMELINE below; it never changes a result or an
 * algorithm.  Every A/B switch and sweep knob of the experiments behind DESIGN_LOG.md is compiled out of this library (csrc/common.h,
 * ab_env()) and lives only in the laboratory build of the same sources (`make -C halo2_amd/csrc ab` -> build/ab/libhalo2_mi355x_ab.so)
 * that the A/B parity tests and bench/tools load; tests/test_abi_and_host.py holds the shipped binary to that.
 *
 * With H2_TIMELINE=1 in the environment every commit stamps the device clock (100 MHz) as its sort, accumulate
 * and reduce stages become runnable; this drains up to `cap` {clock, (stream id << 8) | stage} pairs, stage
 * 1 = sort, 2 = accumulate, 3 = reduce, 4 = done.  Returns the pair count, or -1 when the timeline is off. */
int h2_debug_timeline(unsigned long long *out, unsigned cap);

#ifdef __cplusplus
}
#endif
#endif /* HALO2_MI355X_H */
