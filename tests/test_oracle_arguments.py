"""CPU-only: the oracle's restatements of the permutation and lookup provers (oracle/permutation.py, oracle/lookup.py) checked
against the properties the reference itself asserts under its `sanity-checks` feature (plonk/lookup/prover.rs:331-362,
:629-642) and the ones its verifier relies on -- so that the GPU parity tests compare the device against a checker that has
been checked."""
import random

import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import ipa, lookup as olk, pasta, permutation as operm


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


@pytest.mark.parametrize("seed", range(6))
def test_permute_expression_pair_restatement_properties(seed):
    rnd = random.Random(seed)
    m = fields.MODULUS[h.FP]
    usable = rnd.choice([1, 2, 7, 26, 100])
    table = [rnd.randrange(m) if rnd.random() < 0.5 else rnd.randrange(4) for _ in range(usable)]
    inputs = [rnd.choice(table) for _ in range(usable)]
    a, s = olk.permute_expression_pair(inputs + [123], table + [456], usable)            # rows beyond `usable` are ignored
    assert a == sorted(inputs) and sorted(s) == sorted(table)
    last = None
    for x_, y_ in zip(a, s):                                                             # prover.rs:629-642
        if x_ != y_:
            assert x_ == last
        last = x_
    missing = next(v for v in range(10 ** 6, 10 ** 6 + usable + 2) if v not in table)
    assert olk.permute_expression_pair([missing] + inputs[1:], table, usable) is None


def test_lookup_product_restatement_satisfies_the_row_identity():
    m = fields.MODULUS[h.FP]
    rnd = random.Random(9)
    n, bf = 32, 5
    usable = n - (bf + 1)
    table = [rnd.randrange(m) for _ in range(n)]
    inputs = [rnd.choice(table[:usable]) for _ in range(usable)] + [rnd.randrange(m) for _ in range(n - usable)]
    a, s = olk.permute_expression_pair(inputs, table, usable)
    a += [rnd.randrange(m) for _ in range(bf + 1)]
    s += [rnd.randrange(m) for _ in range(bf + 1)]
    beta, gamma = rnd.randrange(m), rnd.randrange(m)
    blind_rows = [rnd.randrange(m) for _ in range(bf)]
    z = olk.product(inputs, table, a, s, beta, gamma, blind_rows, m)
    assert len(z) == n and z[0] == 1 and z[usable] == 1 and z[n - bf:] == blind_rows   # prover.rs:334, :361
    for i in range(usable):                                                              # prover.rs:338-356
        left = z[i + 1] * (beta + a[i]) % m * (gamma + s[i]) % m
        right = z[i] * (inputs[i] + beta) % m * (table[i] + gamma) % m
        assert left == right
    # a lookup that does not hold breaks the last value
    bad = list(inputs)
    bad[3] = next(v for v in range(1, 100) if v not in table)
    z_bad = olk.product(bad, table, a, s, beta, gamma, blind_rows, m)
    assert z_bad[usable] != 1


@pytest.mark.parametrize("cs_degree", [3, 4, 6])
def test_permutation_restatement_ends_at_one_iff_the_copy_constraints_hold(cs_degree):
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    k, bf, n_cols = 5, 5, 4
    n = 1 << k
    usable = n - (bf + 1)
    rnd = random.Random(cs_degree)
    dom = pasta.EvaluationDomain(cs_degree, k, m)
    cols = [[rnd.randrange(m) for _ in range(n)] for _ in range(n_cols)]
    mapping = [[(c, r) for r in range(n)] for c in range(n_cols)]
    cells = [(c, r) for c in range(n_cols) for r in range(usable)]
    rnd.shuffle(cells)
    for i in range(0, 60, 3):                                                            # twenty 3-cycles of equal cells
        group = cells[i:i + 3]
        v = rnd.randrange(m)
        for j, (c, r) in enumerate(group):
            cols[c][r] = v
            mapping[c][r] = group[(j + 1) % 3]
    sigmas = operm.build_sigma(mapping, dom)
    # sigma is a permutation of the delta^c omega^r labels
    labels = sorted(pow(operm.DELTA[m], c, m) * pow(dom.omega, r, m) % m for c in range(n_cols) for r in range(n))
    assert sorted(v for col in sigmas for v in col) == labels
    g = co.generate_bases(curve, 5, n)
    w = co.generate_bases(curve, 6, 1)[0]
    beta, gamma = rnd.randrange(m), rnd.randrange(m)
    sets = operm.commit(curve, dom, g, w, cs_degree, bf, cols, sigmas, beta, gamma, _rng(sf, 1), ipa.Transcript(curve))
    assert len(sets) == -(-n_cols // (cs_degree - 2))
    assert sets[0][0][0] == 1 and sets[-1][0][usable] == 1
    for (z_prev, _), (z_next, _) in zip(sets, sets[1:]):                                 # each set starts where the previous one ended
        assert z_next[0] == z_prev[usable]
    broken = [list(c) for c in cols]
    c0, r0 = cells[0]
    broken[c0][r0] = (broken[c0][r0] + 1) % m
    sets_b = operm.commit(curve, dom, g, w, cs_degree, bf, broken, sigmas, beta, gamma, _rng(sf, 1), ipa.Transcript(curve))
    assert sets_b[-1][0][usable] != 1


@pytest.mark.parametrize("field", [h.FP, h.FQ])
def test_delta_is_the_documented_constant(field):
    """ff::PrimeField::DELTA = MULTIPLICATIVE_GENERATOR^(2^S) (the `ff` crate's definition; no numeric value is pinned anywhere in
    the reference tree).  The book asks for a T-th root of unity, p - 1 = 2^S T (book/src/design/proving-system/permutation.md:148):
    delta^T = 1, delta generates more than any small-index subgroup, and the product's and the oracle's constants agree."""
    m = fields.MODULUS[field]
    d = fields.delta(field)
    assert d == operm.DELTA[m] == pow(5, 1 << 32, m)
    t = (m - 1) >> 32
    assert t % 2 == 1 and (m - 1) == t << 32
    assert pow(d, t, m) == 1 and d != 1
    for q in (3, 5, 7, 11, 13, 17, 19, 23):
        if t % q == 0:
            assert pow(d, t // q, m) != 1
    # the labels delta^c omega^r of different columns never collide (what the permutation argument needs): delta is outside the 2^S-torsion
    assert pow(d, 1 << 32, m) != 1
