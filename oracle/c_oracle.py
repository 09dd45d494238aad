"""ctypes binding of oracle/libh2oracle.so (TEST INFRASTRUCTURE ONLY).

Arrays are numpy uint64: field elements (n,4), affine points (n,8), Jacobian (12,),
Montgomery form (what a Rust `Vec<Fp>` holds in memory) unless stated otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import pasta

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libh2oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "h2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:    # several ranks may arrive here at once
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libh2oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        u64p = C.POINTER(C.c_uint64)
        sig = {
            "orc_set_threads": ([C.c_int], None),
            "orc_get_threads": ([], C.c_int),
            "orc_f_mul": ([C.c_int, u64p, u64p, u64p], None),
            "orc_f_add": ([C.c_int, u64p, u64p, u64p], None),
            "orc_f_sub": ([C.c_int, u64p, u64p, u64p], None),
            "orc_f_inv": ([C.c_int, u64p, u64p], None),
            "orc_to_mont": ([C.c_int, u64p, C.c_size_t], None),
            "orc_from_mont": ([C.c_int, u64p, C.c_size_t], None),
            "orc_point_to_affine": ([C.c_int, u64p, u64p], None),
            "orc_point_add": ([C.c_int, u64p, u64p, u64p], None),
            "orc_point_mul": ([C.c_int, u64p, u64p, u64p], None),
            "orc_point_on_curve": ([C.c_int, u64p], C.c_int),
            "orc_batch_to_affine": ([C.c_int, u64p, u64p, C.c_size_t], None),
            "orc_window_bits": ([C.c_size_t], C.c_int),
            "orc_best_multiexp": ([C.c_int, u64p, u64p, C.c_size_t, u64p], C.c_int),
            "orc_commit": ([C.c_int, u64p, u64p, u64p, u64p, C.c_size_t, u64p], C.c_int),
            "orc_best_fft": ([C.c_int, u64p, u64p, C.c_uint], C.c_int),
            "orc_ifft": ([C.c_int, u64p, u64p, C.c_uint, u64p], C.c_int),
            "orc_distribute_powers_zeta": ([C.c_int, u64p, C.c_size_t, u64p], None),
            "orc_coeff_to_extended": ([C.c_int, u64p, C.c_uint, C.c_uint, u64p, u64p, u64p], C.c_int),
            "orc_extended_to_coeff": ([C.c_int, u64p, C.c_uint, u64p, u64p, u64p, u64p], C.c_int),
            "orc_divide_by_vanishing_poly": ([C.c_int, u64p, C.c_uint, u64p, C.c_size_t], None),
            "orc_eval_polynomial": ([C.c_int, u64p, C.c_size_t, u64p, u64p], None),
            "orc_inner_product": ([C.c_int, u64p, u64p, C.c_size_t, u64p], None),
            "orc_kate_division": ([C.c_int, u64p, C.c_size_t, u64p, u64p], None),
            "orc_powers": ([C.c_int, u64p, C.c_size_t, u64p], None),
            "orc_scale_add": ([C.c_int, u64p, u64p, u64p, C.c_size_t], None),
            "orc_batch_invert": ([C.c_int, u64p, C.c_size_t], None),
            "orc_grand_product": ([C.c_int, u64p, C.c_size_t, u64p, u64p], None),
            "orc_random_field": ([C.c_int, C.c_uint64, u64p, C.c_size_t], None),
            "orc_generate_bases": ([C.c_int, u64p, C.c_uint64, u64p, C.c_size_t], None),
            "orc_msm_naive": ([C.c_int, u64p, u64p, C.c_size_t, u64p], C.c_int),
            "orc_generator_collapse": ([C.c_int, u64p, C.c_size_t, u64p], None),
            "orc_lagrange_basis": ([C.c_int, u64p, C.c_uint, u64p], C.c_int),
            "orc_fold_scalars": ([C.c_int, u64p, C.c_size_t, u64p], None),
        }
        for name, (args, res) in sig.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = res
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


# --- int <-> limb helpers ------------------------------------------------------
def ints_to_limbs(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a: np.ndarray) -> list[int]:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]


def to_mont(field: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_to_mont(field, _p(a), a.size // 4)
    return a


def from_mont(field: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_from_mont(field, _p(a), a.size // 4)
    return a


def field_of_curve(curve: int, which: str) -> int:
    """field id of the base / scalar field of curve (0 Pallas, 1 Vesta)."""
    if which == "base":
        return 1 if curve else 0
    return 0 if curve else 1


def points_to_mont(curve: int, pts) -> np.ndarray:
    """list of affine (x,y) int tuples / None -> (n,8) Montgomery limbs (identity = zeros)."""
    flat = []
    for p in pts:
        flat += [0, 0] if p is None else [p[0], p[1]]
    a = ints_to_limbs(flat)
    return to_mont(field_of_curve(curve, "base"), a).reshape(-1, 8)


def affine_to_ints(curve: int, xy: np.ndarray):
    """(8,) Montgomery affine -> canonical (x, y) tuple or None for the identity."""
    v = limbs_to_ints(from_mont(field_of_curve(curve, "base"), xy.reshape(2, 4)))
    return None if v == [0, 0] else (v[0], v[1])


def jac_to_affine_ints(curve: int, xyz: np.ndarray):
    out = np.zeros(8, dtype=np.uint64)
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
    lib().orc_point_to_affine(curve, _p(out), _p(xyz))
    return affine_to_ints(curve, out)


# --- the restated hot path -----------------------------------------------------
def best_multiexp(curve: int, scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
    n = scalars.shape[0]
    assert bases.shape[0] == n
    out = np.zeros(12, dtype=np.uint64)
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    rc = lib().orc_best_multiexp(curve, _p(s), _p(b), n, _p(out))
    assert rc == 0
    return out


def msm_naive(curve: int, scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    lib().orc_msm_naive(curve, _p(s), _p(b), s.shape[0], _p(out))
    return out


def commit(curve, g, w, poly, blind) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    w = np.ascontiguousarray(w, dtype=np.uint64)
    poly = np.ascontiguousarray(poly, dtype=np.uint64)
    blind = np.ascontiguousarray(blind, dtype=np.uint64)
    rc = lib().orc_commit(curve, _p(g), _p(w), _p(poly), _p(blind), poly.shape[0], _p(out))
    assert rc == 0
    return out


def best_fft(field: int, a: np.ndarray, omega: np.ndarray, log_n: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    assert a.shape[0] == 1 << log_n
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    rc = lib().orc_best_fft(field, _p(a), _p(omega), log_n)
    assert rc == 0
    return a


def ifft(field, a, omega_inv, log_n, divisor) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    rc = lib().orc_ifft(field, _p(a), _p(np.ascontiguousarray(omega_inv)), log_n,
                        _p(np.ascontiguousarray(divisor)))
    assert rc == 0
    return a


def coeff_to_extended(field, a, k, ext_k, g_coset, g_coset_inv, extended_omega) -> np.ndarray:
    ext = np.zeros((1 << ext_k, 4), dtype=np.uint64)
    ext[: 1 << k] = a
    rc = lib().orc_coeff_to_extended(field, _p(ext), k, ext_k, _p(np.ascontiguousarray(g_coset)),
                                     _p(np.ascontiguousarray(g_coset_inv)),
                                     _p(np.ascontiguousarray(extended_omega)))
    assert rc == 0
    return ext


def extended_to_coeff(field, a_ext, ext_k, g_coset, g_coset_inv, ext_omega_inv, ext_divisor) -> np.ndarray:
    a = np.ascontiguousarray(a_ext, dtype=np.uint64).copy()
    rc = lib().orc_extended_to_coeff(field, _p(a), ext_k, _p(np.ascontiguousarray(g_coset)),
                                     _p(np.ascontiguousarray(g_coset_inv)),
                                     _p(np.ascontiguousarray(ext_omega_inv)),
                                     _p(np.ascontiguousarray(ext_divisor)))
    assert rc == 0
    return a


def divide_by_vanishing_poly(field, a_ext, ext_k, t_evals) -> np.ndarray:
    a = np.ascontiguousarray(a_ext, dtype=np.uint64).copy()
    t = np.ascontiguousarray(t_evals, dtype=np.uint64)
    lib().orc_divide_by_vanishing_poly(field, _p(a), ext_k, _p(t), t.shape[0])
    return a


# --- synthetic inputs ------------------------------------------------------------
def random_field(field: int, seed: int, n: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_random_field(field, seed, _p(out), n)
    return out


# curve generators used for synthetic bases: Pallas (-1, 2) is pinned at
# halo2_proofs/src/poly/commitment/msm.rs:181; (-1, 2) is also on Vesta (same equation).
def generator(curve: int) -> np.ndarray:
    m = pasta.CURVES[curve][0]
    return points_to_mont(curve, [(m - 1, 2)])[0]


def generate_bases(curve: int, seed: int, n: int) -> np.ndarray:
    out = np.empty((n, 8), dtype=np.uint64)
    g = generator(curve)
    lib().orc_generate_bases(curve, _p(g), seed, _p(out), n)
    return out


def generator_collapse(curve: int, g: np.ndarray, challenge: np.ndarray) -> np.ndarray:
    """parallel_generator_collapse (poly/commitment/prover.rs:154-166): returns the collapsed first half."""
    g = np.ascontiguousarray(g, dtype=np.uint64).copy()
    half = g.shape[0] // 2
    lib().orc_generator_collapse(curve, _p(g), half, _p(np.ascontiguousarray(challenge, dtype=np.uint64)))
    return g[:half]


def fold_scalars(field: int, a: np.ndarray, factor: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    half = a.shape[0] // 2
    lib().orc_fold_scalars(field, _p(a), half, _p(np.ascontiguousarray(factor, dtype=np.uint64)))
    return a[:half]


def _fe(v) -> np.ndarray:
    return np.ascontiguousarray(v, dtype=np.uint64).reshape(4)


def eval_polynomial(field: int, poly: np.ndarray, point) -> np.ndarray:
    poly = np.ascontiguousarray(poly, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_eval_polynomial(field, _p(poly), poly.shape[0], _p(_fe(point)), _p(out))
    return out


def inner_product(field: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    assert a.shape == b.shape
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_inner_product(field, _p(a), _p(b), a.shape[0], _p(out))
    return out


def kate_division(field: int, a: np.ndarray, point) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    q = np.zeros((a.shape[0] - 1, 4), dtype=np.uint64)
    lib().orc_kate_division(field, _p(a), a.shape[0], _p(_fe(point)), _p(q))
    return q


def powers(field: int, x, n: int) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_powers(field, _p(_fe(x)), n, _p(out))
    return out


def scale_add(field: int, a: np.ndarray, x, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    b = np.ascontiguousarray(b, dtype=np.uint64)
    lib().orc_scale_add(field, _p(a), _p(_fe(x)), _p(b), a.shape[0])
    return a


def batch_invert(field: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_batch_invert(field, _p(a), a.shape[0])
    return a


def grand_product(field: int, m: np.ndarray, n: int, init) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.uint64)
    z = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_grand_product(field, _p(m), n, _p(_fe(init)), _p(z))
    return z


def lagrange_basis(curve: int, g: np.ndarray, k: int) -> np.ndarray:
    """g_lagrange of Params::new (poly/commitment.rs:77-100) from g."""
    g = np.ascontiguousarray(g, dtype=np.uint64)
    out = np.zeros_like(g)
    assert lib().orc_lagrange_basis(curve, _p(g), k, _p(out)) == 0
    return out
