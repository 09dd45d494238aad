"""`best_multiexp` / `best_fft` over the C ABI -- the two free functions every MSM and FFT of the
reference funnels through (halo2_proofs/src/arithmetic.rs:143, :192).

Host arrays are numpy uint64: scalars / field vectors (n, 4), affine bases (n, 8), Montgomery limbs --
the bytes a Rust `Vec<Fp>` / `Vec<EqAffine>` holds.  Device arrays are torch CUDA tensors of the same
shapes (dtype int64 or uint64); then the call is asynchronous on torch's current stream and returns a
device tensor.  torch is plumbing here (device memory + streams), nothing more.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib, u64p


def _np(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError(f"expected shape (n, {cols}), got {a.shape}")
    return a


def _p(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def best_multiexp(coeffs, bases, curve: int, form: int = FORM_MONTGOMERY, affine: bool = False):
    """sum_i coeffs[i] * bases[i]  (arithmetic.rs:143).  Raises ValueError when the lengths differ,
    where the reference panics (`assert_eq!`, :144).  Returns Jacobian (12,) limbs like `C::Curve`, or
    affine (8,) when `affine`."""
    out_kind = OUT_AFFINE if affine else OUT_JACOBIAN
    out_len = 8 if affine else 12
    if _is_torch(coeffs):
        import torch
        if coeffs.shape[0] != bases.shape[0]:
            raise ValueError("best_multiexp: coeffs and bases differ in length")
        assert coeffs.is_cuda and bases.is_cuda and coeffs.is_contiguous() and bases.is_contiguous()
        out = torch.empty(out_len, dtype=coeffs.dtype, device=coeffs.device)
        rc = lib().h2_msm_device(curve, coeffs.data_ptr(), bases.data_ptr(), coeffs.shape[0], form, out_kind,
                                 out.data_ptr(), _stream_ptr())
        check(rc, "h2_msm_device")
        return out
    coeffs, bases = _np(coeffs, 4), _np(bases, 8)
    if coeffs.shape[0] != bases.shape[0]:
        raise ValueError("best_multiexp: coeffs and bases differ in length")
    out = np.zeros(out_len, dtype=np.uint64)
    check(lib().h2_msm(curve, _p(coeffs), _p(bases), coeffs.shape[0], form, out_kind, _p(out)), "h2_msm")
    return out


def best_fft(a, omega, log_n: int, field: int, form: int = FORM_MONTGOMERY):
    """In-place radix-2 FFT, natural order in and out (arithmetic.rs:192).  Raises ValueError unless
    len(a) == 1 << log_n (the reference asserts, :205).  `omega`: (4,) limbs in the same form as `a`."""
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    if a.shape[0] != (1 << log_n):
        raise ValueError("best_fft: len(a) != 1 << log_n")
    if _is_torch(a):
        assert a.is_cuda and a.is_contiguous()
        check(lib().h2_ntt_device(field, a.data_ptr(), log_n, _p(omega), form, _stream_ptr()), "h2_ntt_device")
        return a
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.ndim == 2 and a.shape[1] == 4):
        raise ValueError("best_fft: `a` must be a C-contiguous (n, 4) uint64 array (transformed in place)")
    check(lib().h2_ntt(field, _p(a), log_n, _p(omega), form), "h2_ntt")
    return a


def points_sum(points_xyz, curve: int) -> np.ndarray:
    """Sum of Jacobian points (count, 12): the local step after all-gathering per-GPU partial MSM results."""
    pts = _np(np.asarray(points_xyz).reshape(-1, 12), 12)
    out = np.zeros(12, dtype=np.uint64)
    check(lib().h2_points_sum(curve, _p(pts), pts.shape[0], _p(out)), "h2_points_sum")
    return out


def msm_window_bits(n: int) -> int:
    return lib().h2_msm_window_bits(n)


def parallel_generator_collapse(g, challenge, curve: int, form: int = FORM_MONTGOMERY):
    """`parallel_generator_collapse` (poly/commitment/prover.rs:154-166): g[i] <- g[i] + [challenge] * g[half + i] for the
    first half of `g` (2 * half affine points), normalised to affine.  Returns the collapsed half (the reference
    truncates, :137).  numpy in -> numpy out; torch CUDA tensor in -> in place, a view of the first half out."""
    challenge = np.ascontiguousarray(challenge, dtype=np.uint64).reshape(4)
    if g.shape[0] % 2:
        raise ValueError("parallel_generator_collapse: odd length")
    half = g.shape[0] // 2
    if _is_torch(g):
        assert g.is_cuda and g.is_contiguous()
        check(lib().h2_generator_collapse_device(curve, g.data_ptr(), half, _p(challenge), form, _stream_ptr()),
              "h2_generator_collapse_device")
        return g[:half]
    g = _np(g, 8).copy()
    check(lib().h2_generator_collapse(curve, _p(g), half, _p(challenge), form), "h2_generator_collapse")
    return g[:half]


def fold_scalars(a, factor, field: int, form: int = FORM_MONTGOMERY):
    """The p' / b collapse of an IPA round (poly/commitment/prover.rs:128-131): a[i] += a[half + i] * factor."""
    factor = np.ascontiguousarray(factor, dtype=np.uint64).reshape(4)
    if a.shape[0] % 2:
        raise ValueError("fold_scalars: odd length")
    half = a.shape[0] // 2
    if _is_torch(a):
        assert a.is_cuda and a.is_contiguous()
        check(lib().h2_fold_scalars_device(field, a.data_ptr(), half, _p(factor), form, _stream_ptr()), "h2_fold_scalars_device")
        return a[:half]
    a = _np(a, 4).copy()
    check(lib().h2_fold_scalars(field, _p(a), half, _p(factor), form), "h2_fold_scalars")
    return a[:half]
