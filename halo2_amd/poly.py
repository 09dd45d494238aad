"""`Polynomial<F, B>` and its basis markers (halo2_proofs/src/poly.rs:30-57): a vector of field elements tagged with the
basis it is written in.  In the reference the tag is a phantom type and mixing bases does not compile; here it is a runtime
tag and the `EvaluationDomain` / `Params` entry points raise `TypeError` for the wrong basis.  The values are the same
(n, 4) uint64 Montgomery-limb array (numpy, host) or torch CUDA tensor the C ABI takes -- wrapping adds no copy."""
from __future__ import annotations


class Basis:
    name = "?"

    def __repr__(self):
        return self.name


class _Coeff(Basis):
    """poly.rs:36-38: coefficients."""
    name = "Coeff"


class _LagrangeCoeff(Basis):
    """poly.rs:41-43: coefficients of the Lagrange basis polynomials (evaluations over the domain)."""
    name = "LagrangeCoeff"


class _ExtendedLagrangeCoeff(Basis):
    """poly.rs:47-49: evaluations over the extended (coset) domain."""
    name = "ExtendedLagrangeCoeff"


Coeff, LagrangeCoeff, ExtendedLagrangeCoeff = _Coeff(), _LagrangeCoeff(), _ExtendedLagrangeCoeff()


class Polynomial:
    """poly.rs:53-57."""
    __slots__ = ("values", "basis")

    def __init__(self, values, basis: Basis):
        if not isinstance(basis, Basis):
            raise TypeError("Polynomial: basis must be Coeff, LagrangeCoeff or ExtendedLagrangeCoeff")
        if getattr(values, "ndim", 0) != 2 or values.shape[1] != 4:
            raise ValueError("Polynomial: values must be an (n, 4) limb array")
        self.values, self.basis = values, basis

    def __len__(self):                       # poly.rs:99-103 num_coeffs / len
        return self.values.shape[0]

    num_coeffs = __len__

    @property
    def shape(self):
        return self.values.shape

    def __getitem__(self, i):                # Index / IndexMut (poly.rs:59-92)
        return self.values[i]

    def __setitem__(self, i, v):
        self.values[i] = v

    def __repr__(self):
        return f"Polynomial<{self.basis}>(len={len(self)})"


def unwrap(a, want: Basis, what: str):
    """The raw array of `a`, checking its basis when it carries one (raw arrays are taken on trust, as before)."""
    if isinstance(a, Polynomial):
        if a.basis is not want:
            raise TypeError(f"{what}: expected a polynomial in the {want} basis, got {a.basis}")
        return a.values, True
    return a, False


def rewrap(values, basis: Basis, tagged: bool):
    return Polynomial(values, basis) if tagged else values
