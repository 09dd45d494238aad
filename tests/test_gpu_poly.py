"""Parity of the polynomial helper kernels (halo2_amd/csrc/poly.hip, through the C ABI) against the oracle's
sequential restatements: bit-exact at every index.  Sizes straddle the 2048-element tile and the 256-tile carry
level of the scans.  Runs only on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 7, 8, 9, 255, 2047, 2048, 2049, 5000, 70001, (1 << 19) + 3]


def _vec(field, seed, n):
    return co.random_field(field, seed, n)


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("n", [0] + SIZES)
def test_eval_and_inner_product(field, n):
    a, b = _vec(field, 11, max(n, 1))[:n], _vec(field, 12, max(n, 1))[:n]
    x = _vec(field, 13, 1)[0]
    assert np.array_equal(h.eval_polynomial(a, x, field), co.eval_polynomial(field, a, x))
    assert np.array_equal(h.compute_inner_product(a, b, field), co.inner_product(field, a, b))


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("n", SIZES)
def test_kate_division_powers_scale_add(field, n):
    a, b = _vec(field, 21, n), _vec(field, 22, n)
    x = _vec(field, 23, 1)[0]
    assert np.array_equal(h.kate_division(a, x, field), co.kate_division(field, a, x))
    assert np.array_equal(h.powers(x, n, field), co.powers(field, x, n))
    assert np.array_equal(h.scale_add(a, x, b, field), co.scale_add(field, a, x, b))


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("n", SIZES)
def test_batch_invert_and_grand_product(field, n):
    a = _vec(field, 31, n)
    a[::5] = 0                                       # BatchInvert leaves zeros alone
    assert np.array_equal(h.batch_invert(a, field), co.batch_invert(field, a))
    m = _vec(field, 32, n)
    init = _vec(field, 33, 1)[0]
    assert np.array_equal(h.grand_product(m, n, init, field), co.grand_product(field, m, n, init))
    if n > 1:                                        # exactly n - 1 factors is enough
        assert np.array_equal(h.grand_product(m[: n - 1].copy(), n, init, field), co.grand_product(field, m, n, init))


def test_canonical_form_matches_montgomery():
    field, n = h.FP, 4099
    a, b = _vec(field, 41, n), _vec(field, 42, n)
    x = _vec(field, 43, 1)[0]
    can = lambda v: co.from_mont(field, v)
    C = h.FORM_CANONICAL
    assert np.array_equal(h.eval_polynomial(can(a), can(x), field, C), can(co.eval_polynomial(field, a, x)))
    assert np.array_equal(h.compute_inner_product(can(a), can(b), field, C), can(co.inner_product(field, a, b)))
    assert np.array_equal(h.kate_division(can(a), can(x), field, C), can(co.kate_division(field, a, x)))
    assert np.array_equal(h.powers(can(x), n, field, C), can(co.powers(field, x, n)))
    assert np.array_equal(h.scale_add(can(a), can(x), can(b), field, C), can(co.scale_add(field, a, x, b)))
    assert np.array_equal(h.batch_invert(can(a), field, C), can(co.batch_invert(field, a)))
    assert np.array_equal(h.grand_product(can(b), n, can(x), field, C), can(co.grand_product(field, b, n, x)))


def test_device_resident_helpers_and_properties_2_20():
    """Device-pointer variants on torch's stream at the full column size, checked through size-independent identities:
    kate_division inverts multiplication by (X - b); z[n-1] * m[n-1] equals the product taken in another order;
    inverse of inverse is the identity."""
    import torch
    field, n = h.FP, 1 << 20
    m = fields.MODULUS[field]
    dev = torch.device("cuda:0")
    a = _vec(field, 51, n)
    x = _vec(field, 52, 1)[0]
    d_a = torch.from_numpy(a.view(np.int64)).to(dev)
    # a(X) = q(X) (X - x) + a(x):  evaluate both sides at a second point y
    y = _vec(field, 53, 1)[0]
    d_q = h.kate_division(d_a, x, field)
    ev = lambda v: co.limbs_to_ints(co.from_mont(field, v.cpu().numpy().view(np.uint64) if hasattr(v, "cpu") else v))[0]
    a_x, a_y, q_y = ev(h.eval_polynomial(d_a, x, field)), ev(h.eval_polynomial(d_a, y, field)), ev(h.eval_polynomial(d_q, y, field))
    xi, yi = ev(x), ev(y)
    assert a_y == (q_y * (yi - xi) + a_x) % m
    assert a_x == ev(co.eval_polynomial(field, a, x))
    # powers + inner product = evaluation
    d_b = h.powers(x, n, field, device=dev)
    assert ev(h.compute_inner_product(d_a, d_b, field)) == a_x
    # grand product vs pairwise tree on the host
    d_z = h.grand_product(d_a, n, fields.scalar_limbs(1, field, True), field)
    z_last = ev(d_z[n - 1])
    ints = np.array(co.limbs_to_ints(co.from_mont(field, a[: n - 1])), dtype=object)
    while ints.shape[0] > 1:
        if ints.shape[0] % 2:
            ints = np.append(ints, 1)
        ints = (ints[0::2] * ints[1::2]) % m
    assert z_last == int(ints[0])
    # inverse twice
    d_i = d_a.clone()
    h.batch_invert(d_i, field)
    assert ev(h.compute_inner_product(d_a[:1000].contiguous(), d_i[:1000].contiguous(), field)) == 1000 % m
    h.batch_invert(d_i, field)
    assert torch.equal(d_i, d_a)
    # scale_add in place
    d_c = d_a.clone()
    h.scale_add(d_c, x, d_b, field)
    assert np.array_equal(d_c[:4096].cpu().numpy().view(np.uint64), co.scale_add(field, a[:4096], x, co.powers(field, x, 4096)))


def test_bad_arguments():
    field = h.FP
    a = _vec(field, 61, 8)
    with pytest.raises(ValueError):
        h.compute_inner_product(a, a[:4], field)
    with pytest.raises(ValueError):
        h.kate_division(a[:0], a[0], field)
    with pytest.raises(ValueError):
        h.grand_product(a[:3], 8, a[0], field)
    with pytest.raises(ValueError):
        h.eval_polynomial(a, a[0], 7)          # unknown field id -> H2_ERR_ARGS


def test_helpers_on_concurrent_streams():
    """Calls on different streams own separate scratch: interleaved reductions / scans / evaluations on three streams give
    the single-stream results."""
    import torch
    from halo2_amd.evaluator import LAGRANGE, Ast, new_evaluator
    field, n = h.FP, 1 << 16
    dev = torch.device("cuda:0")
    dom = h.EvaluationDomain(3, 16, field)
    cols = [_vec(field, 200 + i, n) for i in range(3)]
    xs = [_vec(field, 210 + i, 1)[0] for i in range(3)]
    d = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    want = [(co.eval_polynomial(field, c, x), co.inner_product(field, c, cols[0]), co.kate_division(field, c, x)) for c, x in zip(cols, xs)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    got = [None] * 3
    evs = []
    for rep in range(4):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                ev = new_evaluator(LAGRANGE)
                leaf = ev.register_poly(d[i])
                e = ev.evaluate(Ast.of(leaf) * 3 + Ast.linear(5), dom)
                got[i] = (h.eval_polynomial(d[i], xs[i], field), h.compute_inner_product(d[i], d[0], field), h.kate_division(d[i], xs[i], field), e)
    torch.cuda.synchronize()
    m = dom.m
    for i in range(3):
        a, b, c, e = got[i]
        assert np.array_equal(a.cpu().numpy().view(np.uint64), want[i][0])
        assert np.array_equal(b.cpu().numpy().view(np.uint64), want[i][1])
        assert np.array_equal(c.cpu().numpy().view(np.uint64), want[i][2])
        vals = co.limbs_to_ints(co.from_mont(field, cols[i][:4]))
        lin = [5 * pow(dom.omega, j, m) % m for j in range(4)]
        assert fields.from_limbs(e[:4].cpu().numpy().view(np.uint64), field, True) == [(3 * v + l) % m for v, l in zip(vals, lin)]
