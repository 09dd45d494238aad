"""The device-resident lookup argument (halo2_amd/lookup.py) on a toy circuit -- a two-column tuple lookup whose first input
expression is a product of two columns: permuted columns, grand product, blinds and transcript bytes against the integer
restatement (oracle/lookup.py); its five constraints carried through the device vanishing argument, with h(x) (x^n - 1)
checked against the VERIFIER's formula (plonk/lookup/verifier.rs:96-170) from opened evaluations; every query through the
device multi-point opening, accepted by the restated verifier; an unsatisfied lookup raises ConstraintSystemFailure.
Runs only on a real MI355X (`-m gpu`)."""
import random

import numpy as np
import pytest
import torch

import halo2_amd as h
from halo2_amd import fields
from halo2_amd._lib import ConstraintSystemFailure
from halo2_amd.evaluator import EXTENDED, LAGRANGE, Ast, new_evaluator
from halo2_amd.lookup import Argument
from halo2_amd.multiopen import ProverQuery, create_proof as multiopen_create_proof
from halo2_amd.transcript import Blake2bWrite
from halo2_amd.vanishing import Argument as VanishingArgument
from oracle import c_oracle as co
from oracle import ipa, lookup as olk, multiopen as om, pasta, vanishing as ov

pytestmark = pytest.mark.gpu


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


def _setup(k, seed, break_row=None):
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n, bf = 1 << k, 5
    usable = n - (bf + 1)
    rnd = random.Random(seed)
    rows = [(rnd.randrange(m), rnd.randrange(m)) for _ in range(max(2, usable // 3))]
    table = [rows[i] if i < len(rows) else rows[0] for i in range(usable)] + [(rnd.randrange(m), rnd.randrange(m)) for _ in range(n - usable)]
    picks = [rnd.choice(rows) for _ in range(usable)] + [(rnd.randrange(m), rnd.randrange(m)) for _ in range(n - usable)]
    q = [1] * usable + [rnd.randrange(m) for _ in range(n - usable)]
    a0, a1 = [p_[0] for p_ in picks], [p_[1] for p_ in picks]
    if break_row is not None:
        a1[break_row] = (a1[break_row] + 1) % m
    t0, t1 = [t_[0] for t_ in table], [t_[1] for t_ in table]
    return curve, sf, m, n, bf, usable, [a0, a1, t0, t1, q]


INPUTS = [lambda c: Ast.of(c[0]) * Ast.of(c[4]), lambda c: Ast.of(c[1])]
TABLES = [lambda c: Ast.of(c[2]), lambda c: Ast.of(c[3])]


@pytest.mark.parametrize("k", [5, 8])
def test_lookup_argument_end_to_end(k):
    curve, sf, m, n, bf, usable, cols = _setup(k, 40 + k)
    cs_degree = 5                                     # z(wX) (a' + beta)(s' + gamma) active and z (a0 q theta + a1 + beta)(..) active
    dom = h.EvaluationDomain(cs_degree, k, sf)
    odom = pasta.EvaluationDomain(cs_degree, k, m)
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda ints: fields.to_limbs([v % m for v in ints], sf, True)
    up = lambda ints: torch.from_numpy(L(ints).view(np.int64)).cuda()
    dn = lambda t: I(t.cpu().numpy().view(np.uint64))
    evalp = lambda poly, pt: sum(cf * pow(pt, i, m) for i, cf in enumerate(poly)) % m
    l0 = [1] + [0] * (n - 1)
    l_last = [1 if r == usable else 0 for r in range(n)]
    l_blind = [1 if r > usable else 0 for r in range(n)]
    g = co.generate_bases(curve, 950 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)

    vals, cosets = new_evaluator(LAGRANGE), new_evaluator(EXTENDED)
    d_lag = [up(c) for c in cols]
    d_coeff = [dom.lagrange_to_coeff(t.clone()) for t in d_lag]
    value_leaves = [vals.register_poly(t) for t in d_lag]
    coset_leaves = [cosets.register_poly(dom.coeff_to_extended(t)) for t in d_coeff]
    l_leaves = [cosets.register_poly(dom.coeff_to_extended(dom.lagrange_to_coeff(up(v)))) for v in (l0, l_blind, l_last)]

    # ---- prover on the device, in plonk::create_proof's order (plonk/prover.rs:396-470)
    tr = Blake2bWrite(curve)
    theta = tr.squeeze_challenge()
    arg = Argument(INPUTS, TABLES)
    permuted = arg.commit_permuted(params, dom, bf, vals, cosets, theta, value_leaves, coset_leaves, _rng(sf, 6000), tr)
    beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge()
    committed = permuted.commit_product(params, dom, bf, beta, gamma, cosets, _rng(sf, 6100), tr)
    vcommitted = VanishingArgument.commit(params, dom, _rng(sf, 6200), tr)
    y = tr.squeeze_challenge()
    constructed, exprs = committed.construct(beta, gamma, *l_leaves)
    vconstructed = vcommitted.construct(params, dom, cosets, exprs, y, _rng(sf, 6300), tr)
    x_l = tr.squeeze_challenge_scalar()
    x = fields.from_limbs(x_l.reshape(1, 4), sf, True)[0]
    xn = pow(x, n, m)
    vevaluated = vconstructed.evaluate(x_l, xn, dom, tr)
    evaluated = constructed.evaluate(dom, x, tr)
    prefix = tr.finalize()

    # ---- the same on integers
    pieces = cs_degree - 1

    class Hook(ipa.Transcript):
        def __init__(self, curve):
            super().__init__(curve)
            self.points = 0

        def write_point(self, pt):
            super().write_point(pt)
            self.points += 1
            if self.points == 2:
                assert self.squeeze_challenge() == beta and self.squeeze_challenge() == gamma
            if self.points == 4:
                assert self.squeeze_challenge() == y
            if self.points == 4 + pieces:
                assert self.squeeze_challenge() == x
    ot = Hook(curve)
    assert ot.squeeze_challenge() == theta
    a0, a1, t0, t1, q = cols
    comp_in = [(theta * (x_ * q_) + y_) % m for x_, y_, q_ in zip(a0, a1, q)]
    comp_tb = [(theta * x_ + y_) % m for x_, y_ in zip(t0, t1)]
    assert dn(permuted.compressed_input_expression) == comp_in and dn(permuted.compressed_table_expression) == comp_tb
    pa, ps, pa_blind, ps_blind = olk.commit_permuted(curve, params.g_lagrange, w, bf, comp_in, comp_tb, _rng(sf, 6000), ot)
    assert dn(permuted.permuted_input_expression) == pa and dn(permuted.permuted_table_expression) == ps
    z, z_blind = olk.commit_product(curve, params.g_lagrange, w, bf, comp_in, comp_tb, pa, ps, beta, gamma, m, _rng(sf, 6100), ot)
    assert z[usable] == 1, "the lookup holds, so the product ends at 1"
    col_coeff = [odom.lagrange_to_coeff(c) for c in cols]
    pa_c, ps_c, z_c = (odom.lagrange_to_coeff(v) for v in (pa, ps, z))
    assert dn(constructed.product_poly) == z_c and dn(constructed.permuted_input_poly) == pa_c
    l_coeff = [odom.lagrange_to_coeff(v) for v in (l0, l_blind, l_last)]
    ext = [odom.coeff_to_extended(p_) for p_ in col_coeff + [pa_c, ps_c, z_c] + l_coeff]
    P = lambda i: ("poly", i, 0)
    a_comp = ("add", ("scale", ("mul", P(0), P(4)), theta), P(1))
    s_comp = ("add", ("scale", P(2), theta), P(3))
    trees = olk.constraint_trees(beta, gamma, m, 7, 5, 6, a_comp, s_comp, 8, 9, 10)
    rv, rh = _rng(sf, 6200), _rng(sf, 6300)
    calls = [rv, rv, rh]
    h_poly, h_blind, random_poly, random_blind = ov.prove(curve, odom, g, w, lambda c: calls.pop(0)(c), ot, ext, trees, y, x)
    assert dn(vevaluated.h_poly) == h_poly
    x_inv, x_next = x * odom.omega_inv % m, x * odom.omega % m
    evs = [evalp(z_c, x), evalp(z_c, x_next), evalp(pa_c, x), evalp(pa_c, x_inv), evalp(ps_c, x)]
    for v in evs:
        ot.write_scalar(v)
    assert bytes(ot.out) == prefix

    # ---- h(x) (x^n - 1) against the verifier's formula, from evaluations only
    ce = [evalp(c, x) for c in col_coeff]
    vexprs = olk.verifier_expressions(evs[0], evs[1], evs[2], evs[3], evs[4], (theta * (ce[0] * ce[4]) + ce[1]) % m,
                                      (theta * ce[2] + ce[3]) % m, evalp(l_coeff[0], x), evalp(l_coeff[2], x), evalp(l_coeff[1], x),
                                      beta, gamma, m)
    expected = 0
    for e in vexprs:
        expected = (expected * y + e) % m
    assert evalp(h_poly, x) * (xn - 1) % m == expected

    # ---- every opened polynomial through the device multi-point opening; the restated verifier accepts
    col_blinds = co.random_field(sf, 6400, len(cols))
    queries = [ProverQuery(x_l, t, h.Blind(b)) for t, b in zip(d_coeff, col_blinds)] + evaluated.open(dom, x) + vevaluated.open(x_l)
    tr2 = Blake2bWrite(curve)
    multiopen_create_proof(params, _rng(sf, 6500), tr2, queries)
    comm = lambda coeff, blind_l: co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(coeff), blind_l))
    c_cols = [comm(c, b) for c, b in zip(col_coeff, col_blinds)]
    c_z, c_a, c_s = comm(z_c, L([z_blind])[0]), comm(pa_c, L([pa_blind])[0]), comm(ps_c, L([ps_blind])[0])
    c_h, c_r = comm(h_poly, L([h_blind])[0]), comm(random_poly, L([random_blind])[0])
    vq = [(x, c, evalp(p_, x)) for c, p_ in zip(c_cols, col_coeff)]
    vq += [(x, c_z, evs[0]), (x, c_a, evs[2]), (x, c_s, evs[4]), (x_inv, c_a, evs[3]), (x_next, c_z, evs[1])]
    vq += [(x, c_h, evalp(h_poly, x)), (x, c_r, evalp(random_poly, x))]
    assert om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, tr2.finalize()), vq)
    params.close()


def test_unsatisfied_lookup_is_refused():
    k = 5
    curve, sf, m, n, bf, usable, cols = _setup(k, 77, break_row=3)
    dom = h.EvaluationDomain(5, k, sf)
    up = lambda ints: torch.from_numpy(fields.to_limbs(ints, sf, True).view(np.int64)).cuda()
    g = co.generate_bases(curve, 960, n)
    params = h.Params(curve, k, g, g, g[1], g[2])
    vals, cosets = new_evaluator(LAGRANGE), new_evaluator(EXTENDED)
    d_lag = [up(c) for c in cols]
    value_leaves = [vals.register_poly(t) for t in d_lag]
    coset_leaves = [cosets.register_poly(dom.coeff_to_extended(dom.lagrange_to_coeff(t.clone()))) for t in d_lag]
    with pytest.raises(ConstraintSystemFailure):
        Argument(INPUTS, TABLES).commit_permuted(params, dom, bf, vals, cosets, 12345, value_leaves, coset_leaves, _rng(sf, 1), Blake2bWrite(curve))
    params.close()
