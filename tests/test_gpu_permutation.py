"""The device-resident permutation argument (halo2_amd/permutation.py) on a toy circuit with random copy constraints:
grand products, transcript bytes and opened polynomials against the integer restatement (oracle/permutation.py); its
constraint expressions carried through the device vanishing argument, with h(x) (x^n - 1) checked against the VERIFIER's
formula for those constraints (plonk/permutation/verifier.rs:102-190) from the opened evaluations; and every query carried
through the device multi-point opening and accepted by the restated verifier.  Runs only on a real MI355X (`-m gpu`)."""
import random

import numpy as np
import pytest
import torch

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.evaluator import EXTENDED, new_evaluator
from halo2_amd.multiopen import ProverQuery, create_proof as multiopen_create_proof
from halo2_amd.permutation import Argument, ProvingKey
from halo2_amd.transcript import Blake2bWrite
from halo2_amd.vanishing import Argument as VanishingArgument
from oracle import c_oracle as co
from oracle import ipa, multiopen as om, pasta, permutation as operm, vanishing as ov

pytestmark = pytest.mark.gpu


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


def _toy_circuit(rnd, m, n, n_cols, usable):
    """Random values with random copy constraints among the usable rows: (columns as integer lists, mapping)."""
    cells = [(c, r) for c in range(n_cols) for r in range(usable)]
    rnd.shuffle(cells)
    mapping = [[(c, r) for r in range(n)] for c in range(n_cols)]
    cols = [[rnd.randrange(m) for _ in range(n)] for _ in range(n_cols)]
    pos = 0
    while pos < len(cells) // 2:                                   # half of the usable cells sit in cycles of 2..5 cells
        size = rnd.randint(2, 5)
        group = cells[pos:pos + size]
        pos += size
        val = rnd.randrange(m)
        for i, (c, r) in enumerate(group):
            cols[c][r] = val
            mapping[c][r] = group[(i + 1) % len(group)]
    return cols, mapping


@pytest.mark.parametrize("cs_degree,k", [(3, 5), (4, 5), (5, 7)])
def test_permutation_argument_end_to_end(cs_degree, k):
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n, bf, n_cols = 1 << k, 5, 3
    usable = n - (bf + 1)
    rnd = random.Random(100 * cs_degree + k)
    dom = h.EvaluationDomain(cs_degree, k, sf)
    odom = pasta.EvaluationDomain(cs_degree, k, m)
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda ints: fields.to_limbs([v % m for v in ints], sf, True)
    up = lambda ints: torch.from_numpy(L(ints).view(np.int64)).cuda()
    dn = lambda t: I(t.cpu().numpy().view(np.uint64))
    evalp = lambda poly, pt: sum(cf * pow(pt, i, m) for i, cf in enumerate(poly)) % m

    cols, mapping = _toy_circuit(rnd, m, n, n_cols, usable)
    sigmas = operm.build_sigma(mapping, odom)
    l0 = [1] + [0] * (n - 1)
    l_last = [1 if r == usable else 0 for r in range(n)]
    l_blind = [1 if r > usable else 0 for r in range(n)]

    g = co.generate_bases(curve, 900 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)          # commit_lagrange and commit must agree

    def three_bases(ints):
        lag = up(ints)
        coeff = dom.lagrange_to_coeff(lag.clone())
        return lag, coeff, dom.coeff_to_extended(coeff)
    ev = new_evaluator(EXTENDED)
    d_cols = [three_bases(c) for c in cols]
    d_sig = [three_bases(s) for s in sigmas]
    col_leaves = [ev.register_poly(t[2]) for t in d_cols]
    sig_leaves = [ev.register_poly(t[2]) for t in d_sig]
    l0_leaf, lblind_leaf, llast_leaf = (ev.register_poly(three_bases(v)[2]) for v in (l0, l_blind, l_last))
    pkey = ProvingKey([t[0] for t in d_sig], [t[1] for t in d_sig], sig_leaves)

    # ---- prover on the device, in plonk::create_proof's order (plonk/prover.rs:430-560)
    tr = Blake2bWrite(curve)
    beta, gamma = tr.squeeze_challenge(), tr.squeeze_challenge()
    committed = Argument(n_cols).commit(params, dom, cs_degree, bf, pkey, [t[0] for t in d_cols], beta, gamma, ev, _rng(sf, 5000), tr)
    n_sets = len(committed.sets)
    assert n_sets == -(-n_cols // (cs_degree - 2))
    vcommitted = VanishingArgument.commit(params, dom, _rng(sf, 5100), tr)
    y = tr.squeeze_challenge()
    constructed, exprs = committed.construct(dom, cs_degree, bf, pkey, col_leaves, l0_leaf, lblind_leaf, llast_leaf, beta, gamma)
    assert len(exprs) == 2 + (n_sets - 1) + n_sets
    vconstructed = vcommitted.construct(params, dom, ev, exprs, y, _rng(sf, 5200), tr)
    x_l = tr.squeeze_challenge_scalar()
    x = fields.from_limbs(x_l.reshape(1, 4), sf, True)[0]
    xn = pow(x, n, m)
    vevaluated = vconstructed.evaluate(x_l, xn, dom, tr)
    pkey.evaluate(x_l, sf, tr)
    evaluated = constructed.evaluate(dom, bf, x, tr)
    prefix = tr.finalize()

    # ---- the same on integers
    pieces = cs_degree - 1

    class Hook(ipa.Transcript):                    # squeezes y and x where the prover did
        def __init__(self, curve):
            super().__init__(curve)
            self.points = 0

        def write_point(self, pt):
            super().write_point(pt)
            self.points += 1
            if self.points == n_sets + 1:
                assert self.squeeze_challenge() == y
            if self.points == n_sets + 1 + pieces:
                assert self.squeeze_challenge() == x
    ot = Hook(curve)
    assert ot.squeeze_challenge() == beta and ot.squeeze_challenge() == gamma
    osets = operm.commit(curve, odom, params.g_lagrange, w, cs_degree, bf, cols, sigmas, beta, gamma, _rng(sf, 5000), ot)
    assert osets[-1][0][usable] == 1, "the toy circuit's copy constraints hold, so the last product ends at 1"
    z_coeff = [odom.lagrange_to_coeff(z) for z, _ in osets]
    for s, zc, (_, zb) in zip(constructed.sets, z_coeff, osets):
        assert dn(s.permutation_product_poly) == zc
        assert fields.from_limbs(s.permutation_product_blind.value.reshape(1, 4), sf, True)[0] == zb
    col_coeff = [odom.lagrange_to_coeff(c) for c in cols]
    sig_coeff = [odom.lagrange_to_coeff(s) for s in sigmas]
    l_coeff = [odom.lagrange_to_coeff(v) for v in (l0, l_blind, l_last)]
    ext = [odom.coeff_to_extended(p_) for p_ in col_coeff + sig_coeff + z_coeff + l_coeff]
    trees = operm.constraint_trees(n_sets, n_cols, cs_degree, bf, beta, gamma, m, 0, n_cols, 2 * n_cols, 2 * n_cols + n_sets,
                                   2 * n_cols + n_sets + 1, 2 * n_cols + n_sets + 2)
    rv, rh = _rng(sf, 5100), _rng(sf, 5200)
    calls = [rv, rv, rh]
    h_poly, h_blind, random_poly, random_blind = ov.prove(curve, odom, g, w, lambda c: calls.pop(0)(c), ot, ext, trees, y, x)
    assert dn(vevaluated.h_poly) == h_poly
    for sc in sig_coeff:
        ot.write_scalar(evalp(sc, x))
    x_next, x_last = x * odom.omega % m, x * pow(odom.omega_inv, bf + 1, m) % m
    z_evals = []
    for i, zc in enumerate(z_coeff):
        e = [evalp(zc, x), evalp(zc, x_next), evalp(zc, x_last) if i + 1 < n_sets else None]
        for v in e:
            if v is not None:
                ot.write_scalar(v)
        z_evals.append(tuple(e))
    assert bytes(ot.out) == prefix

    # ---- h(x) (x^n - 1) against the verifier's formula for the permutation constraints, from evaluations only
    vexprs = operm.verifier_expressions(cs_degree, [evalp(c, x) for c in col_coeff], [evalp(s, x) for s in sig_coeff], z_evals,
                                        evalp(l_coeff[0], x), evalp(l_coeff[2], x), evalp(l_coeff[1], x), beta, gamma, x, m)
    expected = 0
    for e in vexprs:
        expected = (expected * y + e) % m
    assert evalp(h_poly, x) * (xn - 1) % m == expected

    # ---- every opened polynomial through the device multi-point opening; the restated verifier accepts
    col_blinds = co.random_field(sf, 5300, n_cols)
    one = fields.scalar_limbs(1, sf, True)
    queries = [ProverQuery(x_l, t[1], h.Blind(b)) for t, b in zip(d_cols, col_blinds)]
    queries += pkey.open(x_l, sf) + evaluated.open(dom, bf, x) + vevaluated.open(x_l)
    tr2 = Blake2bWrite(curve)
    multiopen_create_proof(params, _rng(sf, 5400), tr2, queries)
    comm = lambda coeff, blind_l: co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(coeff), blind_l))
    c_cols = [comm(c, b) for c, b in zip(col_coeff, col_blinds)]
    c_sig = [comm(s, one) for s in sig_coeff]
    c_z = [comm(zc, L([zb])[0]) for zc, (_, zb) in zip(z_coeff, osets)]
    c_h, c_r = comm(h_poly, L([h_blind])[0]), comm(random_poly, L([random_blind])[0])
    vq = [(x, c, evalp(p_, x)) for c, p_ in zip(c_cols, col_coeff)] + [(x, c, evalp(p_, x)) for c, p_ in zip(c_sig, sig_coeff)]
    for c, zc in zip(c_z, z_coeff):
        vq += [(x, c, evalp(zc, x)), (x_next, c, evalp(zc, x_next))]
    for c, zc in reversed(list(zip(c_z, z_coeff))[:-1]):
        vq.append((x_last, c, evalp(zc, x_last)))
    vq += [(x, c_h, evalp(h_poly, x)), (x, c_r, evalp(random_poly, x))]
    assert len(vq) == len(queries)
    assert om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, tr2.finalize()), vq)
    params.close()
