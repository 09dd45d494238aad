"""The device-resident vanishing argument (halo2_amd/vanishing.py) on a satisfied toy circuit: transcript bytes and the
polynomials it opens against the oracle's integer restatement (oracle/vanishing.py), the defining identity
h(x) (x^n - 1) = sum_i y^i gate_i(x), and the two queries carried through the device multi-point opening and accepted by
the restated verifier.  Runs only on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest
import torch

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.evaluator import EXTENDED, Ast, new_evaluator
from halo2_amd.multiopen import create_proof as multiopen_create_proof
from halo2_amd.transcript import Blake2bWrite
from halo2_amd.vanishing import Argument
from oracle import c_oracle as co
from oracle import ipa, multiopen as om, pasta, vanishing as ov

pytestmark = pytest.mark.gpu


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


@pytest.mark.parametrize("curve,k", [(h.VESTA, 5), (h.PALLAS, 4), (h.VESTA, 9)])
def test_vanishing_argument_on_a_satisfied_circuit(curve, k):
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    dom = h.EvaluationDomain(3, k, sf)                                  # degree-3 constraint system: h has 2 pieces
    odom = pasta.EvaluationDomain(3, k, m)
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    up = lambda ints: torch.from_numpy(fields.to_limbs(ints, sf, True).view(np.int64)).cuda()
    # Lagrange columns of a satisfied circuit: c = a * b on every row, d = a shifted by one row
    a = I(co.random_field(sf, 700 + k, n))
    b = I(co.random_field(sf, 701 + k, n))
    c = [x_ * y_ % m for x_, y_ in zip(a, b)]
    d = a[1:] + a[:1]
    cols = [a, b, c, d]
    coeff_i = [odom.lagrange_to_coeff(col) for col in cols]
    ext_i = [odom.coeff_to_extended(co_) for co_ in coeff_i]
    d_coeff = [dom.lagrange_to_coeff(up(col)) for col in cols]
    d_ext = [dom.coeff_to_extended(t) for t in d_coeff]
    for t, want in zip(d_ext, ext_i):
        assert I(t.cpu().numpy().view(np.uint64)) == want
    g = co.generate_bases(curve, 800 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)

    ev = new_evaluator(EXTENDED)
    la, lb, lc, ld = (ev.register_poly(t) for t in d_ext)
    gates = [Ast.of(la) * Ast.of(lb) - Ast.of(lc), Ast.of(la.with_rotation(1)) - Ast.of(ld)]
    trees = [("add", ("mul", ("poly", 0, 0), ("poly", 1, 0)), ("scale", ("poly", 2, 0), m - 1)),
             ("add", ("poly", 0, 1), ("scale", ("poly", 3, 0), m - 1))]

    tr = Blake2bWrite(curve)
    committed = Argument.commit(params, dom, _rng(sf, 4000), tr)
    y = tr.squeeze_challenge()
    constructed = committed.construct(params, dom, ev, gates, y, _rng(sf, 4100), tr)
    assert len(constructed.h_pieces) == dom.quotient_poly_degree == 2
    x_l = tr.squeeze_challenge_scalar()
    x = fields.from_limbs(x_l.reshape(1, 4), sf, True)[0]
    xn = pow(x, n, m)
    evaluated = constructed.evaluate(x_l, xn, dom, tr)

    # the same steps on integers; the restatement squeezes y and x where the prover did
    rc, rh = _rng(sf, 4000), _rng(sf, 4100)
    calls = [rc, rc, rh]                                                # commit draws twice, construct once
    o_rng = lambda count: calls.pop(0)(count)

    class Hook(ipa.Transcript):                                         # squeeze y after the first point, x after the h pieces
        def __init__(self, curve):
            super().__init__(curve)
            self.points = 0

        def write_point(self, pt):
            super().write_point(pt)
            self.points += 1
            if self.points == 1:
                assert self.squeeze_challenge() == y
            if self.points == 1 + 2:
                assert self.squeeze_challenge() == x
    hook = Hook(curve)
    h_poly, h_blind, random_poly, random_blind = ov.prove(curve, odom, g, w, o_rng, hook, ext_i, trees, y, x)
    assert bytes(hook.out) == tr.finalize()
    assert I(evaluated.h_poly.cpu().numpy().view(np.uint64)) == h_poly
    assert fields.from_limbs(evaluated.h_blind.value.reshape(1, 4), sf, True)[0] == h_blind

    # the identity the verifier relies on (vanishing/verifier.rs:94-107): h(x) * (x^n - 1) = y * gate_0(x) + gate_1(x)
    evalp = lambda poly, pt: sum(cf * pow(pt, i, m) for i, cf in enumerate(poly)) % m
    ax, bx, cx, dx = (evalp(p_, x) for p_ in coeff_i)
    a_wx = evalp(coeff_i[0], x * odom.omega % m)
    assert evalp(h_poly, x) * (xn - 1) % m == (y * (ax * bx - cx) + (a_wx - dx)) % m

    # the two queries through the device multi-point opening, accepted by the restated verifier
    queries = evaluated.open(x_l)
    tr2 = Blake2bWrite(curve)
    multiopen_create_proof(params, _rng(sf, 4200), tr2, queries)
    proof = tr2.finalize()
    h_comm = co.jac_to_affine_ints(curve, co.commit(curve, g, w, fields.to_limbs(h_poly, sf, True), fields.scalar_limbs(h_blind, sf, True)))
    r_comm = co.jac_to_affine_ints(curve, co.commit(curve, g, w, fields.to_limbs(random_poly, sf, True),
                                                    fields.scalar_limbs(random_blind, sf, True)))
    vq = [(x, h_comm, evalp(h_poly, x)), (x, r_comm, evalp(random_poly, x))]
    assert om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof), vq)
    params.close()
