"""TEST INFRASTRUCTURE ONLY: restatement of `poly::Evaluator::evaluate` (halo2_proofs/src/poly/evaluator.rs:129-228 with
the BasisOps of :330-614) on Python integers, one whole vector per node as the reference's `recurse` does (chunking does
not change any value).  Trees are nested tuples:
    ("poly", index, rotation) ("add", a, b) ("mul", a, b) ("scale", a, s) ("distribute", [terms], base) ("linear", s) ("constant", s)"""
from __future__ import annotations

COEFF, LAGRANGE, EXTENDED = 0, 1, 2


def evaluate(tree, polys, basis: int, m: int, k: int, extended_k: int, omega: int, extended_omega: int, zeta: int):
    n = len(polys[0])

    def rec(t):
        kind = t[0]
        if kind == "poly":                                     # get_chunk_of_rotated (:228-237, :599-608), poly.rs:198-217
            _, index, rot = t
            if basis == COEFF:
                assert rot == 0, "Can't rotate polynomials in the standard basis"
            step = 1 if basis != EXTENDED else 1 << (extended_k - k)
            return [polys[index][(i + rot * step) % n] for i in range(n)]      # rotate_left(rot) / rotate_right(-rot)
        if kind == "add":
            return [(a + b) % m for a, b in zip(rec(t[1]), rec(t[2]))]
        if kind == "mul":
            assert basis != COEFF                              # Mul: Lagrange and extended bases (:370-418)
            return [a * b % m for a, b in zip(rec(t[1]), rec(t[2]))]
        if kind == "scale":
            return [a * t[2] % m for a in rec(t[1])]
        if kind == "distribute":                               # :186-196
            acc = [0] * n
            for term in t[1]:
                acc = [(a * t[2] + b) % m for a, b in zip(acc, rec(term))]
            return acc
        if kind == "linear":                                   # :160-181, :209-226, :583-598
            s = t[1] % m
            if basis == COEFF:
                return [s if i == 1 else 0 for i in range(n)]
            w = omega if basis == LAGRANGE else extended_omega
            scale = s if basis == LAGRANGE else s * zeta % m
            return [pow(w, i, m) * scale % m for i in range(n)]
        if kind == "constant":                                 # :147-158, :200-207, :574-581
            s = t[1] % m
            return [s if (basis != COEFF or i == 0) else 0 for i in range(n)]
        raise ValueError(kind)
    return rec(tree)
