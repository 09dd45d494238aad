"""`poly::Evaluator` on the device (halo2_amd/evaluator.py -> h2_evaluate_device) against the oracle's restatement of
evaluator.rs:129-228, all three bases: seeded random trees (rotations, scalings, products, DistributePowers, linear and
constant terms), the reference's short-chunk regression cases (evaluator.rs:625-662), rejected trees, and a gate-shaped
expression at the full extended size checked through an algebraic identity.  Runs only on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.evaluator import COEFF, EXTENDED, LAGRANGE, Ast, new_evaluator
from oracle import c_oracle as co
from oracle import evaluator as oev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _to_dev(ints, field):
    return torch.from_numpy(fields.to_limbs(ints, field, True).view(np.int64)).cuda()


def _from_dev(t, field):
    return fields.from_limbs(t.cpu().numpy().view(np.uint64), field, True)


def _random_tree(rng, leaves, basis, m, depth):
    """Returns (Ast, oracle tuple)."""
    r = rng.integers(0, 100)
    if depth == 0 or r < 25:
        pick = rng.integers(0, 10)
        if pick < 7:
            i = int(rng.integers(0, len(leaves)))
            rot = 0 if basis == COEFF else int(rng.integers(-3, 4))
            return Ast.of(leaves[i].with_rotation(rot)), ("poly", i, rot)
        s = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % m
        return (Ast.constant(s), ("constant", s)) if pick < 9 else (Ast.linear(s), ("linear", s))
    a, ta = _random_tree(rng, leaves, basis, m, depth - 1)
    if r < 45:
        b, tb = _random_tree(rng, leaves, basis, m, depth - 1)
        return a + b, ("add", ta, tb)
    if r < 60 and basis != COEFF:
        b, tb = _random_tree(rng, leaves, basis, m, depth - 1)
        return a * b, ("mul", ta, tb)
    if r < 75:
        s = int(rng.integers(1, 1 << 62)) ** 3 % m
        return a * s, ("scale", ta, s)
    if r < 85:
        return -a, ("scale", ta, m - 1)
    terms = [(a, ta)] + [_random_tree(rng, leaves, basis, m, depth - 1) for _ in range(int(rng.integers(0, 3)))]
    base = int(rng.integers(1, 1 << 62)) ** 2 % m
    return Ast.distribute_powers([t[0] for t in terms], base), ("distribute", [t[1] for t in terms], base)


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("basis", [COEFF, LAGRANGE, EXTENDED])
def test_random_trees_match_oracle(field, basis):
    k = 5
    dom = h.EvaluationDomain(4, k, field)
    m = dom.m
    n = dom.extended_len() if basis == EXTENDED else dom.n
    rng = np.random.default_rng(1234 + 10 * basis + field)
    polys = [[int(x) for x in co.limbs_to_ints(co.from_mont(field, co.random_field(field, 50 + j, n)))] for j in range(4)]
    ev = new_evaluator(basis)
    leaves = [ev.register_poly(_to_dev(p, field)) for p in polys]
    for _ in range(12):
        ast, tree = _random_tree(rng, leaves, basis, m, 4)
        got = _from_dev(ev.evaluate(ast, dom), field)
        want = oev.evaluate(tree, polys, basis, m, dom.k, dom.extended_k, dom.omega, dom.extended_omega, dom.g_coset)
        assert got == want, tree


@pytest.mark.parametrize("basis", [COEFF, LAGRANGE, EXTENDED])
def test_short_chunk_regression_cases(basis):
    """evaluator.rs:625-662: constant and linear terms over an empty polynomial, tiny k."""
    for k in (1, 2, 3):
        dom = h.EvaluationDomain(1, k, h.FP)
        n = dom.extended_len() if basis == EXTENDED else dom.n
        ev = new_evaluator(basis)
        ev.register_poly(_to_dev([0] * n, h.FP))
        for ast, tree in ((Ast.constant(0), ("constant", 0)), (Ast.linear(0), ("linear", 0)), (Ast.constant(7), ("constant", 7)),
                          (Ast.linear(5), ("linear", 5))):
            want = oev.evaluate(tree, [[0] * n], basis, dom.m, dom.k, dom.extended_k, dom.omega, dom.extended_omega, dom.g_coset)
            assert _from_dev(ev.evaluate(ast, dom), h.FP) == want


def test_rejected_trees():
    dom = h.EvaluationDomain(3, 4, h.FP)
    ev_c, ev_l, ev_e = new_evaluator(COEFF), new_evaluator(LAGRANGE), new_evaluator(EXTENDED)
    a = ev_c.register_poly(_to_dev(range(dom.n), h.FP))
    b = ev_l.register_poly(_to_dev(range(dom.n), h.FP))
    c = ev_e.register_poly(_to_dev(range(dom.extended_len()), h.FP))
    with pytest.raises(ValueError):
        ev_c.evaluate(Ast.of(a.with_rotation(1)), dom)               # "Can't rotate polynomials in the standard basis"
    with pytest.raises(ValueError):
        ev_c.evaluate(Ast.of(a) * Ast.of(a), dom)                    # Mul exists for the two Lagrange bases only (evaluator.rs:370-418)
    with pytest.raises(ValueError):
        ev_l.evaluate(Ast.of(c), dom)                                # leaf of another evaluator
    with pytest.raises(ValueError):
        new_evaluator(LAGRANGE).evaluate(Ast.constant(1), dom)       # nothing registered
    def full(depth):                                                 # a complete binary tree needs depth + 1 stack slots
        return Ast.of(c) if depth == 0 else full(depth - 1) + full(depth - 1)
    with pytest.raises(ValueError):
        ev_e.evaluate(full(9), dom)                                  # 10 slots > the kernel's 9
    assert _from_dev(ev_e.evaluate(full(8), dom), h.FP) == [256 * i % dom.m for i in range(dom.extended_len())]
    for nested in ("left", "right"):                                 # chains run in 2 slots whichever way they lean
        chain = Ast.of(c)
        for _ in range(40):
            chain = chain + Ast.of(c) if nested == "left" else Ast.of(c) + chain
        assert _from_dev(ev_e.evaluate(chain, dom), h.FP) == [41 * i % dom.m for i in range(dom.extended_len())]


def test_gate_shaped_expression_2_21():
    """A custom-gate-like tree at the simple-example size (k = 20, extended 2^21): (a * b - c) * q with rotations, then
    the same value assembled from separately evaluated pieces -- a size-independent consistency check -- and spot values
    against big-integer arithmetic."""
    field = h.FP
    dom = h.EvaluationDomain(3, 20, field)
    m, n = dom.m, dom.extended_len()
    cols = [co.random_field(field, 70 + j, n) for j in range(4)]
    d = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    ev = new_evaluator(EXTENDED)
    a, b, c, q = (ev.register_poly(t) for t in d)
    y = 0x1234567890ABCDEF1234567
    gate = (Ast.of(a) * Ast.of(b.with_rotation(1)) - Ast.of(c.with_rotation(-1))) * Ast.of(q)
    full = ev.evaluate(Ast.distribute_powers([gate, Ast.of(a) + Ast.linear(3), Ast.one()], y), dom)
    g_only = ev.evaluate(gate, dom)
    rest = ev.evaluate(Ast.distribute_powers([Ast.of(a) + Ast.linear(3), Ast.one()], y), dom)
    ev2 = new_evaluator(EXTENDED)
    lg, lr = ev2.register_poly(g_only), ev2.register_poly(rest)
    again = ev2.evaluate(Ast.of(lg) * (y * y % m) + Ast.of(lr), dom)
    assert torch.equal(full, again)
    ints = lambda arr, idx: co.limbs_to_ints(co.from_mont(field, arr[idx:idx + 1]))[0]
    step = 1 << (dom.extended_k - dom.k)
    host = full.cpu().numpy().view(np.uint64)
    for i in (0, 1, 12345, n - 1):
        av, bv, cv, qv = ints(cols[0], i), ints(cols[1], (i + step) % n), ints(cols[2], (i - step) % n), ints(cols[3], i)
        gv = (av * bv - cv) * qv % m
        lin = 3 * dom.g_coset * pow(dom.extended_omega, i, m) % m
        want = ((gv * y + av + lin) * y + 1) % m
        assert ints(host, i) == want
