"""TEST INFRASTRUCTURE ONLY: restatement of the vanishing argument's prover (halo2_proofs/src/plonk/vanishing/prover.rs:37-190)
on Python integers (oracle/pasta.py's EvaluationDomain, oracle/evaluator.py) and the C oracle's commits.  Nothing here is
imported by the product; tests compare the device-resident prover's transcript bytes and polynomials with it."""
from __future__ import annotations

import numpy as np

from . import c_oracle as co
from . import evaluator as oev
from . import pasta as o


def prove(curve, dom: o.EvaluationDomain, g, w, rng, transcript, ext_polys, expressions, y: int, x: int):
    """commit (:38-61) -> construct (:65-123) -> evaluate (:127-156).  ext_polys: integer lists over the extended domain;
    expressions: oracle/evaluator.py trees, highest power of y first.  Returns (h_poly ints, h_blind, random_poly ints,
    random_blind) -- the two polynomials `open` (:160-177) queries at x."""
    sf = co.field_of_curve(curve, "scalar")
    m = dom.m
    n = dom.n
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    random_poly_l = rng(n).copy()
    random_blind_l = rng(1)[0].copy()
    transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g, w, random_poly_l, random_blind_l)))
    h_ext = oev.evaluate(("distribute", list(expressions), y), ext_polys, oev.EXTENDED, m, dom.k, dom.extended_k, dom.omega,
                         dom.extended_omega, dom.g_coset)
    h_coeff = dom.extended_to_coeff(dom.divide_by_vanishing_poly(h_ext))
    pieces = [h_coeff[i * n:(i + 1) * n] for i in range(len(h_coeff) // n)]
    blinds_l = rng(len(pieces)).copy()
    for piece, b in zip(pieces, blinds_l):
        transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(piece), b)))
    xn = pow(x, n, m)
    h_poly, h_blind = [0] * n, 0
    for piece, b in zip(reversed(pieces), reversed(I(blinds_l))):
        h_poly = [(a * xn + c) % m for a, c in zip(h_poly, piece)]
        h_blind = (h_blind * xn + b) % m
    random_poly = I(random_poly_l)
    transcript.write_scalar(sum(c * pow(x, i, m) for i, c in enumerate(random_poly)) % m)
    return h_poly, h_blind, random_poly, I(random_blind_l)[0]
