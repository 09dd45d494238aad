"""`EvaluationDomain` (halo2_proofs/src/poly/domain.rs:20-383) over the C ABI.

Domain constants are computed on the host exactly as `EvaluationDomain::new` does (:40-146); the
transforms run on the MI355X.  Vectors are (n, 4) uint64 Montgomery limbs (numpy, host) or torch CUDA
tensors (device; asynchronous on the current stream)."""
from __future__ import annotations

import numpy as np

from . import fields
from ._lib import FORM_MONTGOMERY, check, lib
from .arithmetic import _is_torch, _p, _stream_ptr
from .poly import Coeff, ExtendedLagrangeCoeff, LagrangeCoeff, Polynomial, rewrap, unwrap


class EvaluationDomain:
    def __init__(self, j: int, k: int, field: int):
        """j = cs.degree(), k = log2 n  (domain.rs:40)."""
        m = fields.MODULUS[field]
        self.field, self.m, self.k = field, m, k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        extended_k = k
        while (1 << extended_k) < self.n * self.quotient_poly_degree:
            extended_k += 1
        if extended_k > fields.S:
            raise ValueError("extended_k exceeds the field's 2-adicity")   # assert!, domain.rs:56
        self.extended_k = extended_k
        w = fields.root_of_unity(field)
        for _ in range(extended_k, fields.S):
            w = w * w % m
        self.extended_omega = w
        for _ in range(k, extended_k):
            w = w * w % m
        self.omega = w
        self.omega_inv = pow(self.omega, -1, m)
        self.extended_omega_inv = pow(self.extended_omega, -1, m)
        self.g_coset = fields.zeta(field)
        self.g_coset_inv = self.g_coset * self.g_coset % m
        self.ifft_divisor = pow(1 << k, -1, m)
        self.extended_ifft_divisor = pow(1 << extended_k, -1, m)
        self.barycentric_weight = pow(self.n, -1, m)
        orig = pow(self.g_coset, self.n, m)
        step = pow(self.extended_omega, self.n, m)
        t, cur = [], orig
        while True:
            t.append(cur)
            cur = cur * step % m
            if cur == orig:
                break
        assert len(t) == 1 << (extended_k - k)
        self.t_evaluations = [pow((v - 1) % m, -1, m) for v in t]

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def _c(self, v: int) -> np.ndarray:
        return fields.scalar_limbs(v, self.field, True)

    # -- domain.rs:227-237
    def lagrange_to_coeff(self, a):
        """Polynomial<LagrangeCoeff> -> Polynomial<Coeff> (raw limb arrays pass through untagged)."""
        a, tagged = unwrap(a, LagrangeCoeff, "lagrange_to_coeff")
        if a.shape[0] != self.n:
            raise ValueError("lagrange_to_coeff: wrong length")
        return rewrap(self._ifft(a, self.omega_inv, self.k, self.ifft_divisor), Coeff, tagged)

    def lagrange_to_coeff_batch(self, columns):
        """lagrange_to_coeff over the independent columns of a phase (device tensors, in place) in one call."""
        for a in columns:
            if a.shape[0] != self.n:
                raise ValueError("lagrange_to_coeff: wrong length")
            assert a.is_cuda and a.is_contiguous()
        if columns:
            import ctypes as C
            arr = (C.c_void_p * len(columns))(*[a.data_ptr() for a in columns])
            check(lib().h2_ifft_batch_device(self.field, arr, len(columns), self.k, _p(self._c(self.omega_inv)),
                                             _p(self._c(self.ifft_divisor)), FORM_MONTGOMERY, _stream_ptr()), "h2_ifft_batch_device")
        return columns

    def _ifft(self, a, omega_inv, log_n, divisor):
        """EvaluationDomain::ifft, domain.rs:375-383 (scale fused into the last NTT pass)."""
        if _is_torch(a):
            check(lib().h2_ifft_device(self.field, a.data_ptr(), log_n, _p(self._c(omega_inv)), _p(self._c(divisor)),
                                       FORM_MONTGOMERY, _stream_ptr()), "h2_ifft_device")
            return a
        a = np.ascontiguousarray(a, dtype=np.uint64)
        check(lib().h2_ifft(self.field, _p(a), log_n, _p(self._c(omega_inv)), _p(self._c(divisor)), FORM_MONTGOMERY),
              "h2_ifft")
        return a

    # -- domain.rs:241-255
    def coeff_to_extended(self, a):
        """Polynomial<Coeff> -> Polynomial<ExtendedLagrangeCoeff>."""
        a, tagged = unwrap(a, Coeff, "coeff_to_extended")
        return rewrap(self._coeff_to_extended(a), ExtendedLagrangeCoeff, tagged)

    def _coeff_to_extended(self, a):
        if a.shape[0] != self.n:
            raise ValueError("coeff_to_extended: wrong length")
        args = (_p(self._c(self.g_coset)), _p(self._c(self.g_coset_inv)), _p(self._c(self.extended_omega)))
        if _is_torch(a):
            import torch
            out = torch.empty((self.extended_len(), 4), dtype=a.dtype, device=a.device)
            check(lib().h2_coeff_to_extended_device(self.field, a.data_ptr(), out.data_ptr(), self.k, self.extended_k,
                                                    *args, FORM_MONTGOMERY, _stream_ptr()), "h2_coeff_to_extended_device")
            return out
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.empty((self.extended_len(), 4), dtype=np.uint64)
        check(lib().h2_coeff_to_extended(self.field, _p(a), _p(out), self.k, self.extended_k, *args, FORM_MONTGOMERY),
              "h2_coeff_to_extended")
        return out

    # -- domain.rs:303-325
    def extended_to_coeff(self, a):
        """Polynomial<ExtendedLagrangeCoeff> -> the n * (degree - 1) coefficients (a plain vector in the reference too, :303)."""
        a, _ = unwrap(a, ExtendedLagrangeCoeff, "extended_to_coeff")
        if a.shape[0] != self.extended_len():
            raise ValueError("extended_to_coeff: wrong length")
        args = (_p(self._c(self.g_coset)), _p(self._c(self.g_coset_inv)), _p(self._c(self.extended_omega_inv)),
                _p(self._c(self.extended_ifft_divisor)))
        keep = self.n * self.quotient_poly_degree
        if _is_torch(a):
            check(lib().h2_extended_to_coeff_device(self.field, a.data_ptr(), self.extended_k, *args, FORM_MONTGOMERY,
                                                    _stream_ptr()), "h2_extended_to_coeff_device")
            return a[:keep]
        a = np.ascontiguousarray(a, dtype=np.uint64)
        check(lib().h2_extended_to_coeff(self.field, _p(a), self.extended_k, *args, FORM_MONTGOMERY),
              "h2_extended_to_coeff")
        return a[:keep]

    # -- domain.rs:329-348
    def divide_by_vanishing_poly(self, a):
        a, tagged = unwrap(a, ExtendedLagrangeCoeff, "divide_by_vanishing_poly")
        return rewrap(self._divide_by_vanishing_poly(a), ExtendedLagrangeCoeff, tagged)

    def _divide_by_vanishing_poly(self, a):
        if a.shape[0] != self.extended_len():
            raise ValueError("divide_by_vanishing_poly: wrong length")
        t = fields.to_limbs(self.t_evaluations, self.field, True)
        if _is_torch(a):
            check(lib().h2_divide_by_vanishing_poly_device(self.field, a.data_ptr(), self.extended_k, _p(t), t.shape[0],
                                                           FORM_MONTGOMERY, _stream_ptr()), "h2_divide_by_vanishing_poly_device")
            return a
        a = np.ascontiguousarray(a, dtype=np.uint64)
        check(lib().h2_divide_by_vanishing_poly(self.field, _p(a), self.extended_k, _p(t), t.shape[0], FORM_MONTGOMERY),
              "h2_divide_by_vanishing_poly")
        return a

    # -- the small accessors and rotations (host logic; integers are canonical field values) ----------------------
    def get_omega(self) -> int:                                    # domain.rs:391
        return self.omega

    def get_omega_inv(self) -> int:                                # domain.rs:397
        return self.omega_inv

    def get_extended_omega(self) -> int:                           # domain.rs:402
        return self.extended_omega

    def get_quotient_poly_degree(self) -> int:                     # domain.rs:475
        return self.quotient_poly_degree

    def empty_coeff(self) -> Polynomial:                           # domain.rs:173
        return Polynomial(np.zeros((self.n, 4), dtype=np.uint64), Coeff)

    def empty_lagrange(self) -> Polynomial:                        # domain.rs:181
        return Polynomial(np.zeros((self.n, 4), dtype=np.uint64), LagrangeCoeff)

    def empty_extended(self) -> Polynomial:                        # domain.rs:207
        return Polynomial(np.zeros((self.extended_len(), 4), dtype=np.uint64), ExtendedLagrangeCoeff)

    def constant_lagrange(self, scalar: int) -> Polynomial:        # domain.rs:198
        return Polynomial(np.tile(self._c(scalar), (self.n, 1)), LagrangeCoeff)

    def constant_extended(self, scalar: int) -> Polynomial:        # domain.rs:216
        return Polynomial(np.tile(self._c(scalar), (self.extended_len(), 1)), ExtendedLagrangeCoeff)

    def coeff_from_vec(self, values) -> Polynomial:                # domain.rs:163
        if values.shape[0] != self.n:
            raise ValueError("coeff_from_vec: wrong length")
        return Polynomial(values, Coeff)

    def lagrange_from_vec(self, values) -> Polynomial:             # domain.rs:151
        if values.shape[0] != self.n:
            raise ValueError("lagrange_from_vec: wrong length")
        return Polynomial(values, LagrangeCoeff)

    def rotate_omega(self, value: int, rotation: int) -> int:
        """value * omega^rotation (domain.rs:408-419)."""
        w = self.omega if rotation >= 0 else self.omega_inv
        return value * pow(w, abs(rotation), self.m) % self.m

    def rotate_extended(self, poly, rotation: int):
        """Rotate an extended-domain polynomial by `rotation` rows of the original domain (domain.rs:258-274):
        rotate_left for rotation >= 0, rotate_right otherwise.  numpy or torch CUDA tensor; returns a new array."""
        poly, tagged = unwrap(poly, ExtendedLagrangeCoeff, "rotate_extended")
        if tagged:
            return Polynomial(self.rotate_extended(poly, rotation), ExtendedLagrangeCoeff)
        if poly.shape[0] != self.extended_len():
            raise ValueError("rotate_extended: wrong length")
        shift = (1 << (self.extended_k - self.k)) * abs(rotation)
        shift = -shift if rotation >= 0 else shift                 # roll(-s) == rotate_left(s)
        if _is_torch(poly):
            import torch
            return torch.roll(poly, shift, 0)                      # a device-to-device copy in two pieces
        return np.roll(poly, shift, axis=0)

    def l_i_range(self, x: int, xn: int, rotations) -> list[int]:
        """Evaluations at x of the Lagrange basis polynomials l_i for the given rotations i (domain.rs:447-472)."""
        rotations = list(rotations)
        m = self.m
        results = [(x - self.rotate_omega(1, r)) % m for r in rotations]
        results = [pow(v, -1, m) if v else 0 for v in results]     # batch_invert leaves zeros alone
        common = (xn - 1) * self.barycentric_weight % m
        return [self.rotate_omega(v * common % m, r) for v, r in zip(results, rotations)]
