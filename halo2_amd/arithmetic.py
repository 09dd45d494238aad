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


def best_multiexp_batch(pairs, curve: int, form: int = FORM_MONTGOMERY, affine: bool = False):
    """Several independent `best_multiexp(coeffs, bases)` over device tensors in one call (the L_j / R_j pair of an
    opening-argument round, poly/commitment/prover.rs:107-108); returns one (len, 12|8) device tensor."""
    import torch
    out_len = 8 if affine else 12
    for coeffs, bases in pairs:
        if coeffs.shape[0] != bases.shape[0]:
            raise ValueError("best_multiexp: coeffs and bases differ in length")
        assert coeffs.is_cuda and bases.is_cuda and coeffs.is_contiguous() and bases.is_contiguous()
    if not pairs:
        return None
    out = torch.empty((len(pairs), out_len), dtype=torch.int64, device=pairs[0][0].device)
    arr = C.c_void_p * len(pairs)
    rc = lib().h2_msm_batch_device(curve, arr(*[c.data_ptr() for c, _ in pairs]), arr(*[b.data_ptr() for _, b in pairs]),
                                   (C.c_size_t * len(pairs))(*[c.shape[0] for c, _ in pairs]), len(pairs), form,
                                   OUT_AFFINE if affine else OUT_JACOBIAN, arr(*[out[i].data_ptr() for i in range(len(pairs))]),
                                   _stream_ptr())
    check(rc, "h2_msm_batch_device")
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


def best_fft_batch(columns, omega, log_n: int, field: int, form: int = FORM_MONTGOMERY):
    """`best_fft` over several independent device vectors of one size in one call (the column FFTs of a prover phase,
    plonk/prover.rs:111-117, 322-327); in place, overlapped on internal streams."""
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    for a in columns:
        if a.shape[0] != (1 << log_n):
            raise ValueError("best_fft: len(a) != 1 << log_n")
        assert a.is_cuda and a.is_contiguous()
    if columns:
        arr = (C.c_void_p * len(columns))(*[a.data_ptr() for a in columns])
        check(lib().h2_ntt_batch_device(field, arr, len(columns), log_n, _p(omega), form, _stream_ptr()), "h2_ntt_batch_device")
    return columns


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


def ipa_round_scalars(p_prime, k: int, j: int, challenges, field: int, out_l, out_r, form: int = FORM_MONTGOMERY) -> None:
    """Scalars of round j's L_j / R_j (poly/commitment/prover.rs:107-108) over the ORIGINAL generators (see
    h2_ipa_round_scalars_device): p_prime is the current p' (2^(k-j), 4) CUDA tensor, challenges the j challenges drawn so
    far, out_l / out_r CUDA tensors whose first 2^k rows are written."""
    assert p_prime.is_cuda and p_prime.is_contiguous() and out_l.is_cuda and out_r.is_cuda
    if p_prime.shape[0] != 1 << (k - j) or out_l.shape[0] < 1 << k or out_r.shape[0] < 1 << k or len(challenges) != j:
        raise ValueError("ipa_round_scalars: shapes do not match (k, j)")
    ch = np.ascontiguousarray(np.stack(challenges), dtype=np.uint64).reshape(j, 4) if j else None
    check(lib().h2_ipa_round_scalars_device(field, p_prime.data_ptr(), k, j, _p(ch) if j else None, form, out_l.data_ptr(),
                                            out_r.data_ptr(), _stream_ptr()), "h2_ipa_round_scalars_device")


def sort_field(a, field: int, form: int = FORM_MONTGOMERY):
    """`Vec<F>::sort()` as the lookup prover uses it (plonk/lookup/prover.rs:574): ascending by canonical value.  CUDA tensor,
    in place."""
    assert a.is_cuda and a.is_contiguous()
    check(lib().h2_sort_device(field, a.data_ptr(), a.shape[0], form, _stream_ptr()), "h2_sort_device")
    return a


def permute_expression_pair(input_expression, table_expression, usable_rows: int, field: int, form: int = FORM_MONTGOMERY):
    """`permute_expression_pair` over the usable rows (plonk/lookup/prover.rs:557-623): (A', S') as new CUDA tensors of
    `usable_rows` elements; the caller appends the blinding rows (:624-627).  Raises ConstraintSystemFailure when an input value
    does not occur in the table (:609-611)."""
    import torch
    assert input_expression.is_cuda and table_expression.is_cuda
    if input_expression.shape[0] < usable_rows or table_expression.shape[0] < usable_rows:
        raise ValueError("permute_expression_pair: columns shorter than usable_rows")
    a = torch.empty((usable_rows, 4), dtype=input_expression.dtype, device=input_expression.device)
    s = torch.empty_like(a)
    check(lib().h2_permute_expression_pair_device(field, _dev_vec(input_expression).data_ptr(), _dev_vec(table_expression).data_ptr(),
                                                  usable_rows, form, a.data_ptr(), s.data_ptr(), _stream_ptr()),
          "h2_permute_expression_pair_device")
    return a, s


# ---- polynomial helpers of arithmetic.rs / the opening argument (numpy in -> numpy out; torch CUDA in -> torch out) ----
def _fe(v) -> np.ndarray:
    return np.ascontiguousarray(v, dtype=np.uint64).reshape(4)


def _dev_vec(a):
    assert a.is_cuda and a.is_contiguous() and a.ndim == 2 and a.shape[1] == 4
    return a


def small_multiexp(coeffs, bases, curve: int, form: int = FORM_MONTGOMERY):
    """`small_multiexp` (arithmetic.rs:116-136) -- same sum as best_multiexp; on the device both are one kernel path."""
    return best_multiexp(coeffs, bases, curve, form)


def eval_polynomial(poly, point, field: int, form: int = FORM_MONTGOMERY):
    """`eval_polynomial` (arithmetic.rs:298-303): sum_i poly[i] * point^i -> (4,) limbs."""
    point = _fe(point)
    if _is_torch(poly):
        import torch
        out = torch.empty(4, dtype=poly.dtype, device=poly.device)
        check(lib().h2_eval_polynomial_device(field, _dev_vec(poly).data_ptr(), poly.shape[0], _p(point), form, out.data_ptr(),
                                              _stream_ptr()), "h2_eval_polynomial_device")
        return out
    poly = _np(poly, 4)
    out = np.zeros(4, dtype=np.uint64)
    check(lib().h2_eval_polynomial(field, _p(poly), poly.shape[0], _p(point), form, _p(out)), "h2_eval_polynomial")
    return out


def compute_inner_product(a, b, field: int, form: int = FORM_MONTGOMERY):
    """`compute_inner_product` (arithmetic.rs:308-318).  Raises ValueError on a length mismatch (the reference asserts)."""
    if a.shape[0] != b.shape[0]:
        raise ValueError("compute_inner_product: lengths differ")
    if _is_torch(a):
        import torch
        out = torch.empty(4, dtype=a.dtype, device=a.device)
        check(lib().h2_inner_product_device(field, _dev_vec(a).data_ptr(), _dev_vec(b).data_ptr(), a.shape[0], form, out.data_ptr(),
                                            _stream_ptr()), "h2_inner_product_device")
        return out
    a, b = _np(a, 4), _np(b, 4)
    out = np.zeros(4, dtype=np.uint64)
    check(lib().h2_inner_product(field, _p(a), _p(b), a.shape[0], form, _p(out)), "h2_inner_product")
    return out


def kate_division(a, b, field: int, form: int = FORM_MONTGOMERY):
    """`kate_division` (arithmetic.rs:322-341): a(X) / (X - b) without the remainder; len(a) - 1 coefficients."""
    b = _fe(b)
    if a.shape[0] == 0:
        raise ValueError("kate_division: empty polynomial")
    if _is_torch(a):
        import torch
        out = torch.empty((a.shape[0] - 1, 4), dtype=a.dtype, device=a.device)
        check(lib().h2_kate_division_device(field, _dev_vec(a).data_ptr(), a.shape[0], _p(b), form, out.data_ptr(), _stream_ptr()),
              "h2_kate_division_device")
        return out
    a = _np(a, 4)
    out = np.zeros((a.shape[0] - 1, 4), dtype=np.uint64)
    check(lib().h2_kate_division(field, _p(a), a.shape[0], _p(b), form, _p(out)), "h2_kate_division")
    return out


def powers(x, n: int, field: int, form: int = FORM_MONTGOMERY, device=None):
    """1, x, x^2, ... x^(n-1): the `b` vector of the opening argument (poly/commitment/prover.rs:90-97).
    `device` = a torch device for a device-resident result."""
    x = _fe(x)
    if device is not None:
        import torch
        out = torch.empty((n, 4), dtype=torch.int64, device=device)
        check(lib().h2_powers_device(field, _p(x), n, form, out.data_ptr(), _stream_ptr()), "h2_powers_device")
        return out
    out = np.zeros((n, 4), dtype=np.uint64)
    check(lib().h2_powers(field, _p(x), n, form, _p(out)), "h2_powers")
    return out


def scale_add(a, x, b, field: int, form: int = FORM_MONTGOMERY):
    """a * x + b coefficient-wise (`s_poly * xi + p_poly`, poly/commitment/prover.rs:70).  torch: in place on `a`."""
    x = _fe(x)
    if a.shape[0] != b.shape[0]:
        raise ValueError("scale_add: lengths differ")
    if _is_torch(a):
        check(lib().h2_scale_add_device(field, _dev_vec(a).data_ptr(), _p(x), _dev_vec(b).data_ptr(), a.shape[0], form, _stream_ptr()),
              "h2_scale_add_device")
        return a
    a, b = _np(a, 4).copy(), _np(b, 4)
    check(lib().h2_scale_add(field, _p(a), _p(x), _p(b), a.shape[0], form), "h2_scale_add")
    return a


def batch_invert(a, field: int, form: int = FORM_MONTGOMERY):
    """ff::BatchInvert (plonk/permutation/prover.rs:118): element-wise inverse, zeros stay zero.  torch: in place."""
    if _is_torch(a):
        check(lib().h2_batch_invert_device(field, _dev_vec(a).data_ptr(), a.shape[0], form, _stream_ptr()), "h2_batch_invert_device")
        return a
    a = _np(a, 4).copy()
    check(lib().h2_batch_invert(field, _p(a), a.shape[0], form), "h2_batch_invert")
    return a


def grand_product(m, n: int, init, field: int, form: int = FORM_MONTGOMERY):
    """z[0] = init, z[i] = z[i-1] * m[i-1] for i < n (plonk/permutation/prover.rs:147-153)."""
    init = _fe(init)
    if m.shape[0] < n - 1:
        raise ValueError("grand_product: need at least n - 1 factors")
    if _is_torch(m):
        import torch
        z = torch.empty((n, 4), dtype=m.dtype, device=m.device)
        check(lib().h2_grand_product_device(field, _dev_vec(m).data_ptr(), n, _p(init), form, z.data_ptr(), _stream_ptr()),
              "h2_grand_product_device")
        return z
    m = _np(m, 4)
    z = np.zeros((n, 4), dtype=np.uint64)
    check(lib().h2_grand_product(field, _p(m), n, _p(init), form, _p(z)), "h2_grand_product")
    return z
