"""CPU-only: the host logic of the multi-point opening argument.  (a) `construct_intermediate_sets` of the product
(halo2_amd/multiopen.py) against the oracle's independent restatement (oracle/multiopen.py) on random query patterns and
on the shapes the reference tests (halo2_proofs/src/poly/multiopen.rs:278-474); (b) the oracle's prover and verifier
restatements round-trip, reject a wrong evaluation and reject contradictory queries -- the reference's `test_roundtrip`
and `test_identical_queries`."""
import random

import numpy as np

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.multiopen import construct_intermediate_sets
from oracle import c_oracle as co
from oracle import ipa, multiopen as om


def test_intermediate_sets_match_restatement_on_random_patterns():
    rnd = random.Random(5)
    for trial in range(200):
        n_comm, n_pts = rnd.randint(1, 7), rnd.randint(1, 5)
        pts = [rnd.randrange(1, 1 << 60) for _ in range(n_pts)]
        qs = []
        for _ in range(rnd.randint(1, 14)):
            qs.append((rnd.choice(pts), rnd.randrange(n_comm), rnd.randrange(1 << 30)))
        got = construct_intermediate_sets(qs)
        want = om.construct_intermediate_sets(qs)
        if want is None:
            assert got is None
            continue
        assert got[1] == want[1]
        assert [(d["commitment"], d["set_index"], d["point_indices"], d["evals"]) for d in got[0]] == \
               [(d["key"], d["set_index"], d["point_indices"], d["evals"]) for d in want[0]]
        # every commitment's set holds exactly the points it was queried at, each point set is distinct
        for d in got[0]:
            assert sorted({p for p, c, _ in qs if c == d["commitment"]}) == sorted(got[1][d["set_index"]])
        assert len({tuple(s) for s in got[1]}) == len(got[1])


def test_intermediate_sets_reference_shapes():
    # a, b at x; c at y (multiopen.rs:318-345): two sets in order of first appearance
    x, y = 11, 22
    data, sets = construct_intermediate_sets([(x, "a", None), (x, "b", None), (y, "c", None)])
    assert sets == [[x], [y]]
    assert [(d["commitment"], d["set_index"]) for d in data] == [("a", 0), ("b", 0), ("c", 1)]
    # the same (commitment, point) twice is refused (multiopen.rs:243-249), whatever the evaluations say
    assert construct_intermediate_sets([(x, "a", 1), (x, "b", 2), (x, "b", 3), (y, "c", 4)]) is None
    assert construct_intermediate_sets([(x, "a", None), (x, "a", None)]) is None
    # points are ordered by first appearance, sets by their sorted index tuples' first appearance
    data, sets = construct_intermediate_sets([(y, "a", None), (x, "a", None), (x, "b", None), (y, "b", None), (x, "c", None)])
    assert sets == [[y, x], [x]] and [d["set_index"] for d in data] == [0, 0, 1]


def _setup(curve, k, seed):
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, seed, n)
    w, u = co.generate_bases(curve, seed + 1, 1)[0], co.generate_bases(curve, seed + 2, 1)[0]
    return n, sf, g, w, u


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


def test_oracle_multiopen_roundtrip_and_rejections():
    curve, k = h.VESTA, 4
    n, sf, g, w, u = _setup(curve, k, 90)
    m = fields.MODULUS[sf]
    I = lambda limbs: fields.from_limbs(np.ascontiguousarray(limbs).reshape(1, 4), sf, True)[0]
    ax = fields.to_limbs([10 + i for i in range(n)], sf, True)            # multiopen.rs:293-306
    bx = fields.to_limbs([100 + i for i in range(n)], sf, True)
    cx = fields.to_limbs([100 + i for i in range(n)], sf, True)
    blind = co.random_field(sf, 93, 1)[0]
    a, b, c = (co.jac_to_affine_ints(curve, co.commit(curve, g, w, p, blind)) for p in (ax, bx, cx))
    assert b == c and b is not c                                          # equal values, distinct commitments (objects)
    xl, yl = co.random_field(sf, 94, 2)
    x, y = I(xl), I(yl)
    avx, bvx, cvy = I(co.eval_polynomial(sf, ax, xl)), I(co.eval_polynomial(sf, bx, xl)), I(co.eval_polynomial(sf, cx, yl))
    tr = ipa.Transcript(curve)
    om.create_proof(curve, k, g, w, u, _rng(sf, 500), tr, [(x, ax, blind), (x, bx, blind), (y, cx, blind)])
    proof = bytes(tr.out)
    assert len(proof) == 32 + 2 * 32 + 32 + 64 * k + 64                   # q', two u_i, then the opening argument
    ok = om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof), [(x, a, avx), (x, b, bvx), (y, c, cvy)])
    assert ok
    bad = om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof), [(x, a, avx), (x, b, avx), (y, c, cvy)])
    assert not bad                                                        # multiopen.rs:355-371: "NB: wrong!"
    contradictory = om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof),
                                    [(x, a, avx), (x, b, (bvx + 1) % m), (x, b, bvx), (y, c, cvy)])
    assert not contradictory                                              # multiopen.rs:461-473: Err(OpeningError)


def test_lagrange_interpolate_restatement():
    m = fields.MODULUS[h.FP]
    rnd = random.Random(3)
    for deg in range(1, 6):
        pts = rnd.sample(range(1, 1000), deg)
        evs = [rnd.randrange(m) for _ in pts]
        poly = om.lagrange_interpolate(pts, evs, m)
        assert len(poly) == deg
        for p_, e_ in zip(pts, evs):
            assert sum(c * pow(p_, i, m) for i, c in enumerate(poly)) % m == e_


def test_verifier_host_helpers_match_the_restatement():
    """halo2_amd/verifier.py's pure-host pieces (no GPU needed): lagrange_interpolate against oracle/multiopen.py's and against
    the defining property; compute_b against its product form (poly/commitment/verifier.rs:144-154)."""
    from halo2_amd import verifier as hv
    m = fields.MODULUS[h.FQ]
    rnd = random.Random(11)
    for deg in range(1, 7):
        pts = rnd.sample(range(1, 10 ** 6), deg)
        evs = [rnd.randrange(m) for _ in pts]
        got = hv.lagrange_interpolate(pts, evs, m)
        assert got == om.lagrange_interpolate(pts, evs, m)
        for p_, e_ in zip(pts, evs):
            assert sum(c * pow(p_, i, m) for i, c in enumerate(got)) % m == e_
    u = [rnd.randrange(1, m) for _ in range(6)]
    x = rnd.randrange(m)
    # b = <s, (1, x, x^2, ...)> with s = compute_s(u, 1): the definition compute_b shortcuts
    s = [1]
    for u_j in reversed(u):
        s = s + [v * u_j % m for v in s]
    assert hv.compute_b(x, u, m) == sum(si * pow(x, i, m) for i, si in enumerate(s)) % m
