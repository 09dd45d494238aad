"""Pins the oracle (oracle/pasta.py and oracle/h2_oracle.c) against the reference's own
known-answer fixtures (tests/golden/, extracted by oracle/extract_fixtures.py) and against
each other.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pasta as o

FIELDS = [("fp", 0, o.P), ("fq", 1, o.Q)]


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "poseidon_kat.json")))


@pytest.fixture(scope="module")
def vks(golden_dir):
    return json.load(open(os.path.join(golden_dir, "pinned_vk.json")))


def _c_poseidon(field, state, mds, rcs):
    """Poseidon permutation (halo2_poseidon/src/lib.rs:106-151) on the C field ops, Montgomery form."""
    lib = co.lib()
    st = co.to_mont(field, co.ints_to_limbs(state))
    mds_m = co.to_mont(field, co.ints_to_limbs([v for r in mds for v in r])).reshape(3, 3, 4)
    rc_m = co.to_mont(field, co.ints_to_limbs([v for r in rcs for v in r])).reshape(-1, 3, 4)
    p = co._p

    def mul(a, b):
        r = np.zeros(4, dtype=np.uint64); lib.orc_f_mul(field, p(r), p(np.ascontiguousarray(a)), p(np.ascontiguousarray(b))); return r

    def add(a, b):
        r = np.zeros(4, dtype=np.uint64); lib.orc_f_add(field, p(r), p(np.ascontiguousarray(a)), p(np.ascontiguousarray(b))); return r

    def sbox(x):
        x2 = mul(x, x); return mul(mul(x2, x2), x)

    for r in range(64):
        full = r < 4 or r >= 60
        st = np.stack([add(st[i], rc_m[r, i]) for i in range(3)])
        if full:
            st = np.stack([sbox(st[i]) for i in range(3)])
        else:
            st[0] = sbox(st[0])
        new = []
        for i in range(3):
            acc = np.zeros(4, dtype=np.uint64)
            for j in range(3):
                acc = add(acc, mul(mds_m[i, j], st[j]))
            new.append(acc)
        st = np.stack(new)
    return co.limbs_to_ints(co.from_mont(field, st))


@pytest.mark.parametrize("name,fid,m", FIELDS)
def test_poseidon_kat_python_and_c(kat, name, fid, m):
    rc = [[int(x, 16) for x in r] for r in kat[name]["round_constants"]]
    mds = [[int(x, 16) for x in r] for r in kat[name]["mds"]]
    assert len(kat[name]["permute"]) == 11
    for i, v in enumerate(kat[name]["permute"]):
        st = [int(x, 16) for x in v["initial_state"]]
        want = [int(x, 16) for x in v["final_state"]]
        assert o.poseidon_permute(st, mds, rc, m) == want
        if i < 3:
            assert _c_poseidon(fid, st, mds, rc) == want


def test_pinned_moduli_omega_and_points(vks):
    npts = 0
    for v in vks:
        assert int(v["base_modulus"], 16) == o.Q
        assert int(v["scalar_modulus"], 16) == o.P
        assert o.omega_for(o.P, v["k"]) == int(v["omega"], 16)
        pts = [(int(p["x"], 16), int(p["y"], 16)) for p in v["points"]]
        for pt in pts:
            assert o.on_curve(pt, o.Q)
        mont = co.points_to_mont(1, pts)
        for row in mont:
            assert co.lib().orc_point_on_curve(1, co._p(np.ascontiguousarray(row))) == 1
        npts += len(pts)
    assert npts == 288
    assert {v["k"] for v in vks} == {5, 11}
    # Pallas (-1, 2): halo2_proofs/src/poly/commitment/msm.rs:181
    assert o.on_curve((o.P - 1, 2), o.P)


def test_montgomery_constants():
    for _, fid, m in FIELDS:
        one = co.to_mont(fid, co.ints_to_limbs([1]))
        assert co.limbs_to_ints(one) == [o.R % m]
        rng = o.SplitMix64(7 + fid)
        vals = [rng.field(m) for _ in range(64)] + [0, 1, m - 1]
        mont = co.to_mont(fid, co.ints_to_limbs(vals))
        assert co.limbs_to_ints(mont) == [v * o.R % m for v in vals]
        assert co.limbs_to_ints(co.from_mont(fid, mont)) == vals
        # same PRNG stream in C and Python
        cr = co.random_field(fid, 7 + fid, 64)
        assert co.limbs_to_ints(co.from_mont(fid, cr)) == vals[:64]


@pytest.mark.parametrize("curve", [0, 1])
def test_curve_ops_c_vs_python(curve, vks):
    m = o.CURVES[curve][0]
    rng = o.SplitMix64(99 + curve)
    g = (m - 1, 2)
    pts = [o.ec_mul(rng.field(o.CURVES[curve][1]), g, m) for _ in range(6)]
    mont = co.points_to_mont(curve, pts)
    lib = co.lib()
    for i in range(5):
        a = np.concatenate([mont[i], co.to_mont(co.field_of_curve(curve, "base"), co.ints_to_limbs([1]))[0]])
        b = np.concatenate([mont[i + 1], co.to_mont(co.field_of_curve(curve, "base"), co.ints_to_limbs([1]))[0]])
        out = np.zeros(12, dtype=np.uint64)
        lib.orc_point_add(curve, co._p(out), co._p(np.ascontiguousarray(a)), co._p(np.ascontiguousarray(b)))
        assert co.jac_to_affine_ints(curve, out) == o.ec_add(pts[i], pts[i + 1], m)
        lib.orc_point_add(curve, co._p(out), co._p(np.ascontiguousarray(a)), co._p(np.ascontiguousarray(a)))
        assert co.jac_to_affine_ints(curve, out) == o.ec_add(pts[i], pts[i], m)


SIZES = [0, 1, 3, 4, 31, 32, 33, 255, 257]


@pytest.mark.parametrize("curve", [0, 1])
def test_best_multiexp_c_vs_python_definition(curve):
    """arithmetic.rs:440-458 test_multiexp restated: Pippenger == naive sum, C == Python."""
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    for n in SIZES:
        scal = co.random_field(sf, 1000 + n, n)
        bases = co.generate_bases(curve, 5000 + n, n)
        got = co.jac_to_affine_ints(curve, co.best_multiexp(curve, scal, bases))
        s_int = co.limbs_to_ints(co.from_mont(sf, scal)) if n else []
        b_int = [co.affine_to_ints(curve, bases[i]) for i in range(n)]
        for b in b_int:
            assert o.on_curve(b, bm)
        want = o.msm_naive(s_int, b_int, bm)
        assert got == want, n
        if n <= 33:
            assert o.best_multiexp(s_int, b_int, bm) == want
        assert co.jac_to_affine_ints(curve, co.msm_naive(curve, scal, bases)) == want


def test_window_rule():
    """arithmetic.rs:146-152 and the values in SURVEY.md appendix A.1."""
    for n, c in [(0, 1), (3, 1), (4, 3), (31, 3), (32, 4), (54, 4), (257, 6), (1 << 16, 12),
                 (1 << 19, 14), (1 << 20, 14), ((1 << 20) + 1, 14), (1 << 21, 15), (1 << 22, 16)]:
        assert co.lib().orc_window_bits(n) == c
        assert o.window_bits(n) == c


def test_msm_edge_cases_c():
    """identity bases, duplicate bases, base with its negation, zero / max scalars."""
    curve = 0
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 40
    bases = co.generate_bases(curve, 31337, n)
    b_int = [co.affine_to_ints(curve, bases[i]) for i in range(n)]
    b_int[3] = None
    b_int[5] = b_int[4]
    b_int[7] = o.ec_neg(b_int[6], bm)
    bases = co.points_to_mont(curve, b_int)
    s_int = co.limbs_to_ints(co.from_mont(sf, co.random_field(sf, 4242, n)))
    s_int[0] = 0
    s_int[1] = sm - 1
    s_int[4] = s_int[5] = 12345
    s_int[6] = s_int[7] = 777
    for i in range(20, 40):
        s_int[i] = 1 << (i - 20) * 12
    scal = co.to_mont(sf, co.ints_to_limbs(s_int))
    want = o.msm_naive(s_int, b_int, bm)
    assert co.jac_to_affine_ints(curve, co.best_multiexp(curve, scal, bases)) == want
    # all scalars equal, all bases equal  ->  n*s*B
    s_same = co.to_mont(sf, co.ints_to_limbs([9] * n))
    b_same = co.points_to_mont(curve, [b_int[0]] * n)
    assert co.jac_to_affine_ints(curve, co.best_multiexp(curve, s_same, b_same)) == o.ec_mul(9 * n, b_int[0], bm)


@pytest.mark.parametrize("name,fid,m", FIELDS)
def test_best_fft_c_vs_python(name, fid, m):
    for log_n in [0, 1, 2, 3, 5, 8]:
        n = 1 << log_n
        a = co.random_field(fid, 50 + log_n, n)
        a_int = co.limbs_to_ints(co.from_mont(fid, a))
        omega = o.omega_for(m, log_n)
        om = co.to_mont(fid, co.ints_to_limbs([omega]))[0]
        got = co.limbs_to_ints(co.from_mont(fid, co.best_fft(fid, a, om, log_n)))
        ref = list(a_int)
        o.best_fft(ref, omega, log_n, m)
        assert got == ref
        if log_n <= 5:
            assert ref == o.fft_definition(a_int, omega, m)
    # benches/fft.rs:17 passes an arbitrary (non-root) omega: the output is then NOT the DFT (the radix-2
    # network assumes omega^(n/2) = -1) but it is still a well-defined function of the butterfly network
    # (arithmetic.rs:226-251); the C restatement must reproduce that network exactly.
    a = co.random_field(fid, 77, 16)
    a_int = co.limbs_to_ints(co.from_mont(fid, a))
    w = 0x1234567 % m
    got = co.limbs_to_ints(co.from_mont(fid, co.best_fft(fid, a, co.to_mont(fid, co.ints_to_limbs([w]))[0], 4)))
    ref = list(a_int)
    o.best_fft(ref, w, 4, m)
    assert got == ref
    assert ref != o.fft_definition(a_int, w, m)


def test_fft_thread_paths_agree():
    """serial iterative branch (log_n <= log_threads) and recursive branch give the same output."""
    fid, m = 0, o.P
    a = co.random_field(fid, 3, 1 << 10)
    om = co.to_mont(fid, co.ints_to_limbs([o.omega_for(m, 10)]))[0]
    lib = co.lib()
    outs = []
    for t in (1, 2, 8, 4096):
        lib.orc_set_threads(t)
        outs.append(co.best_fft(fid, a, om, 10))
    lib.orc_set_threads(0)
    for x in outs[1:]:
        assert np.array_equal(outs[0], x)


def test_domain_c_vs_python():
    """EvaluationDomain transforms (poly/domain.rs) incl. test_rotate/test_l_i style checks."""
    fid, m = 0, o.P
    dom = o.EvaluationDomain(3, 6, m)   # cs_degree 3 -> extended_k = 7
    assert dom.extended_k == 7
    mont = lambda v: co.to_mont(fid, co.ints_to_limbs([v]))[0]
    a = co.random_field(fid, 11, dom.n)
    a_int = co.limbs_to_ints(co.from_mont(fid, a))
    coeff = dom.lagrange_to_coeff(a_int)
    c_coeff = co.ifft(fid, a, mont(dom.omega_inv), dom.k, mont(dom.ifft_divisor))
    assert co.limbs_to_ints(co.from_mont(fid, c_coeff)) == coeff
    # iFFT correctness via evaluation at omega^i (domain.rs:500-539 in spirit)
    for i in (0, 1, 17):
        x = pow(dom.omega, i, m)
        assert sum(c * pow(x, j, m) for j, c in enumerate(coeff)) % m == a_int[i]
    ext = dom.coeff_to_extended(coeff)
    c_ext = co.coeff_to_extended(fid, c_coeff, dom.k, dom.extended_k, mont(dom.g_coset), mont(dom.g_coset_inv),
                                 mont(dom.extended_omega))
    assert co.limbs_to_ints(co.from_mont(fid, c_ext)) == ext
    # coset evaluation: ext[j] = poly(zeta * extended_omega^j)
    for j in (0, 5, 100):
        x = dom.g_coset * pow(dom.extended_omega, j, m) % m
        assert sum(c * pow(x, i, m) for i, c in enumerate(coeff)) % m == ext[j]
    back = dom.extended_to_coeff(ext)
    assert back[: dom.n] == coeff and all(v == 0 for v in back[dom.n:])
    c_back = co.extended_to_coeff(fid, c_ext, dom.extended_k, mont(dom.g_coset), mont(dom.g_coset_inv),
                                  mont(dom.extended_omega_inv), mont(dom.extended_ifft_divisor))
    assert co.limbs_to_ints(co.from_mont(fid, c_back))[: dom.n * dom.quotient_poly_degree] == back
    t = co.to_mont(fid, co.ints_to_limbs(dom.t_evaluations))
    c_div = co.divide_by_vanishing_poly(fid, c_ext, dom.extended_k, t)
    assert co.limbs_to_ints(co.from_mont(fid, c_div)) == dom.divide_by_vanishing_poly(ext)


def test_commit_lagrange_equals_commit():
    """poly/commitment.rs:258-302 restated: commit(iFFT(a)) == commit_lagrange(a) where g_lagrange is the
    point-iFFT of g (commitment.rs:77-88)."""
    curve, k = 1, 4
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 1 << k
    g = co.generate_bases(curve, 2024, n)
    g_int = [co.affine_to_ints(curve, g[i]) for i in range(n)]
    w_int = o.ec_mul(424242, (bm - 1, 2), bm)
    dom = o.EvaluationDomain(1, k, sm)
    # g_lagrange[i] = (1/n) sum_j omega^{-ij} g[j]
    g_lag = []
    for i in range(n):
        acc = None
        for j in range(n):
            acc = o.ec_add(acc, o.ec_mul(pow(dom.omega_inv, i * j, sm) * dom.ifft_divisor % sm, g_int[j], bm), bm)
        g_lag.append(acc)
    a = co.random_field(sf, 31, n)
    a_int = co.limbs_to_ints(co.from_mont(sf, a))
    blind = 987654321
    coeff = dom.lagrange_to_coeff(a_int)
    c1 = o.commit(g_int, w_int, coeff, blind, bm)
    c2 = o.commit(g_lag, w_int, a_int, blind, bm)
    assert c1 == c2
    got = co.commit(curve, g, co.points_to_mont(curve, [w_int])[0], co.to_mont(sf, co.ints_to_limbs(coeff)),
                    co.to_mont(sf, co.ints_to_limbs([blind]))[0])
    assert co.jac_to_affine_ints(curve, got) == c1


@pytest.mark.parametrize("name,fid,m", FIELDS)
def test_poly_helpers_c_vs_python_definition(name, fid, m):
    """The C restatements of eval_polynomial / compute_inner_product / kate_division (arithmetic.rs:298-341), the
    powers-of-x vector (poly/commitment/prover.rs:90-97), BatchInvert and the grand product
    (plonk/permutation/prover.rs:118,147-153) against Python big-integer arithmetic of the definitions."""
    rng = o.SplitMix64(0x706F6C79 + fid)
    n = 37
    a = [rng.next() * rng.next() * rng.next() * rng.next() % m for _ in range(n)]
    b = [rng.next() * rng.next() * rng.next() * rng.next() % m for _ in range(n)]
    x = rng.next() * rng.next() * rng.next() % m
    a[5] = 0
    mont = lambda vals: co.to_mont(fid, co.ints_to_limbs(vals))
    ints = lambda arr: co.limbs_to_ints(co.from_mont(fid, arr))
    am, bm, xm = mont(a), mont(b), mont([x])[0]

    assert ints(co.eval_polynomial(fid, am, xm)) == [sum(c * pow(x, i, m) for i, c in enumerate(a)) % m]
    assert ints(co.inner_product(fid, am, bm)) == [sum(u * v for u, v in zip(a, b)) % m]
    assert ints(co.powers(fid, xm, n)) == [pow(x, i, m) for i in range(n)]
    assert ints(co.scale_add(fid, am, xm, bm)) == [(u * x + v) % m for u, v in zip(a, b)]

    # kate_division: q(X) (X - x) + a(x) == a(X)
    q = ints(co.kate_division(fid, am, xm))
    assert len(q) == n - 1
    ev = sum(c * pow(x, i, m) for i, c in enumerate(a)) % m
    back = [(-x * q[0] + ev) % m] + [(q[i - 1] - x * q[i]) % m for i in range(1, n - 1)] + [q[n - 2]]
    assert back == a

    inv = ints(co.batch_invert(fid, am))
    assert inv == [pow(v, -1, m) if v else 0 for v in a]

    init = rng.next() % m
    z = ints(co.grand_product(fid, bm, n, mont([init])[0]))
    want, cur = [], init
    for i in range(n):
        want.append(cur)
        cur = cur * b[i] % m
    assert z == want


def test_point_encoding_roundtrip_and_rejections(vks):
    """oracle to_bytes / from_bytes (Params::write/read, poly/commitment.rs:169-205) on the reference's pinned Vesta
    points: decode(encode(P)) == P, sign bit = parity of y, identity = zeros, invalid encodings rejected."""
    m = o.Q
    pts = [(int(pt["x"], 16), int(pt["y"], 16)) for pt in vks[0]["points"]][:12]      # tests/plonk_api.rs:959-980
    assert len(pts) == 12 and all(o.on_curve(pt, m) for pt in pts)
    for pt in pts:
        enc = o.point_to_bytes(pt, m)
        assert enc[31] >> 7 == pt[1] & 1
        assert o.point_from_bytes(enc, m) == pt
        neg = (pt[0], m - pt[1])
        assert o.point_from_bytes(o.point_to_bytes(neg, m), m) == neg
    assert o.point_to_bytes(None, m) == bytes(32) and o.point_from_bytes(bytes(32), m) is None
    with pytest.raises(ValueError):
        o.point_from_bytes(bytes(31) + b"\x80", m)                     # (0, odd)
    with pytest.raises(ValueError):
        o.point_from_bytes(b"\xff" * 31 + b"\x7f", m)                  # x >= modulus
    raw = o.params_write(2, pts[:4], pts[4:8], pts[8], pts[9], m)
    assert len(raw) == 4 + 32 * 10
    assert o.params_read(raw, m) == (2, pts[:4], pts[4:8], pts[8], pts[9])
