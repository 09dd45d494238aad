"""TEST INFRASTRUCTURE ONLY: restatement of the vanishing argument's prover (halo2_proofs/src/plonk/vanishing/prover.rs:37-190)
on Python integers (oracle/pasta.py's EvaluationDomain, oracle/evaluator.py) and the C oracle's commits.  Nothing here is
imported by the product; tests compare the device-resident prover's transcript bytes and polynomials with it."""
from __future__ import annotations

import numpy as np

from . import c_oracle as co
from . import evaluator as oev
from . import pasta as o


def commit(curve, dom: o.EvaluationDomain, g, w, rng, transcript):
    """Argument::commit (:38-61) -> (random_poly ints, random_blind int)."""
    sf = co.field_of_curve(curve, "scalar")
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    random_poly_l = rng(dom.n).copy()
    random_blind_l = rng(1)[0].copy()
    transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g, w, random_poly_l, random_blind_l)))
    return I(random_poly_l), I(random_blind_l)[0]


def construct(curve, dom: o.EvaluationDomain, g, w, rng, transcript, ext_polys, expressions, y: int):
    """Committed::construct (:65-123) -> (h pieces as integer lists, their blinds)."""
    sf = co.field_of_curve(curve, "scalar")
    m, n = dom.m, dom.n
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    h_ext = oev.evaluate(("distribute", list(expressions), y), ext_polys, oev.EXTENDED, m, dom.k, dom.extended_k, dom.omega,
                         dom.extended_omega, dom.g_coset)
    h_coeff = dom.extended_to_coeff(dom.divide_by_vanishing_poly(h_ext))
    pieces = [h_coeff[i * n:(i + 1) * n] for i in range(len(h_coeff) // n)]
    blinds_l = rng(len(pieces)).copy()
    for piece, b in zip(pieces, blinds_l):
        transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(piece), b)))
    return pieces, I(blinds_l)


def evaluate(dom: o.EvaluationDomain, pieces, blinds, random_poly, x: int, transcript):
    """Constructed::evaluate (:127-156) -> (h_poly ints, h_blind)."""
    m, n = dom.m, dom.n
    xn = pow(x, n, m)
    h_poly, h_blind = [0] * n, 0
    for piece, b in zip(reversed(pieces), reversed(blinds)):
        h_poly = [(a * xn + c) % m for a, c in zip(h_poly, piece)]
        h_blind = (h_blind * xn + b) % m
    transcript.write_scalar(sum(c * pow(x, i, m) for i, c in enumerate(random_poly)) % m)
    return h_poly, h_blind


def prove(curve, dom: o.EvaluationDomain, g, w, rng, transcript, ext_polys, expressions, y: int, x: int):
    """commit -> construct -> evaluate in one go.  ext_polys: integer lists over the extended domain; expressions:
    oracle/evaluator.py trees, highest power of y first.  Returns (h_poly ints, h_blind, random_poly ints, random_blind) -- the
    two polynomials `open` (:160-177) queries at x."""
    random_poly, random_blind = commit(curve, dom, g, w, rng, transcript)
    pieces, blinds = construct(curve, dom, g, w, rng, transcript, ext_polys, expressions, y)
    h_poly, h_blind = evaluate(dom, pieces, blinds, random_poly, x, transcript)
    return h_poly, h_blind, random_poly, random_blind
