"""Group-level parity pinned to values the REFERENCE TREE holds (CPU side; the GPU side is tests/test_gpu_reference_goldens.py).

1. `Params::new(5)` for Vesta + the keygen of the reference's own test circuit reproduce, bit for bit, the verifying key
   pinned at halo2_proofs/tests/plonk_api.rs:958-981 (7 fixed + 12 permutation commitments, fixture
   tests/golden/pinned_vk.json extracted from that file by oracle/extract_fixtures.py).  Every one of those 19 numbers is a
   `commit_lagrange` output: hash-to-curve generators -> point iFFT -> MSM + blind -> to_affine.  Through the Python oracle
   (definition arithmetic) and through the C restatement (`orc_lagrange_basis`, `orc_commit` = `best_multiexp`).
2. Group-law anchors on all 288 reference-pinned Vesta points and Pallas (-1, 2) (poly/commitment/msm.rs:181):
   [q]P = O, [q - 1]P = -P, [2^k]P by doubling == best_multiexp, through oracle/pasta.py and oracle/h2_oracle.c.
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import hash_to_curve as h2c
from oracle import pasta as o
from oracle import plonk_api as pa

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
VKS = json.load(open(os.path.join(GOLDEN, "pinned_vk.json")))
PLONK_API = VKS[0]
assert PLONK_API["source"].endswith("tests/plonk_api.rs")
PINNED = [(int(p["x"], 16), int(p["y"], 16)) for p in PLONK_API["points"]]
PINNED_FIXED, PINNED_PERM = PINNED[:7], PINNED[7:]
VESTA = 1


@pytest.fixture(scope="module")
def params5():
    return pa.params_new("vesta", 5)


def test_hash_to_curve_reproduces_pinned_w():
    """fixed_commitments[0] (plonk_api.rs:959) commits to the never-assigned column `sf` with Blind::default() = 1:
    it IS w = hash_to_curve("Halo2-Parameters")(&[1]) (commitment.rs:102-103).  Exactly one of the six automorphism
    variants of the derived isogeny hits it — the normalised one."""
    assert int(PLONK_API["base_modulus"], 16) == o.Q and int(PLONK_API["scalar_modulus"], 16) == o.P
    iso = h2c.derive_isogeny("vesta")
    hits = [h2c.hash_to_curve("vesta", "Halo2-Parameters", v)(b"\x01") == PINNED_FIXED[0]
            for v in h2c.automorphism_variants(iso, o.Q)]
    assert hits == [True, False, False, False, False, False]
    for cid in ("pallas", "vesta"):                   # the Pallas map is the same construction over Fp
        pt = h2c.hash_to_curve(cid, "z.cash:test")(b"Trans rights now!")      # benches/hashtocurve.rs:15-20 inputs
        assert pt is not None and o.on_curve(pt, h2c.BASE[cid])


def test_params_new_and_keygen_reproduce_pinned_vk_python(params5):
    g, g_lagrange, w, u = params5
    assert w == PINNED_FIXED[0]
    assert all(o.on_curve(p, o.Q) for p in g + g_lagrange + [w, u]) and len(set(g)) == 32
    assert o.omega_for(o.P, 5) == int(PLONK_API["omega"], 16)
    fixed, perm = pa.pinned_commitments(g_lagrange, w, o.Q, o.P)
    assert fixed == PINNED_FIXED
    assert perm == PINNED_PERM
    assert fixed[2] == fixed[3]                      # sa and sb hold the same values (plonk_api.rs:961-962)


def test_pinned_vk_through_c_restatement(params5):
    """Same goldens through oracle/h2_oracle.c: g_lagrange from `orc_lagrange_basis` (the point FFT), commitments from
    `orc_commit` (best_multiexp over g_lagrange || w).  This is the checker every GPU parity test compares against."""
    g, g_lagrange, w, _ = params5
    sf = co.field_of_curve(VESTA, "scalar")
    g_m = co.points_to_mont(VESTA, g)
    gl_m = co.lagrange_basis(VESTA, g_m, 5)
    assert [co.affine_to_ints(VESTA, gl_m[i]) for i in range(32)] == g_lagrange
    w_m = co.points_to_mont(VESTA, [w])[0]
    one = co.to_mont(sf, co.ints_to_limbs([1]))[0]
    fixed, mapping = pa.keygen_columns(o.P)
    om = [pow(o.omega_for(o.P, 5), j, o.P) for j in range(32)]
    sigmas = [[pow(pa.DELTA[o.P], mapping[i][j][0], o.P) * om[mapping[i][j][1]] % o.P for j in range(32)] for i in range(12)]
    got = [co.jac_to_affine_ints(VESTA, co.commit(VESTA, gl_m, w_m, co.to_mont(sf, co.ints_to_limbs(col)), one))
           for col in fixed + sigmas]
    assert got == PINNED
    # and as a bare best_multiexp (arithmetic.rs:143) over g_lagrange || w
    bases = np.concatenate([gl_m, w_m.reshape(1, 8)])
    sc = co.to_mont(sf, co.ints_to_limbs(fixed[pa.SL] + [1]))
    assert co.jac_to_affine_ints(VESTA, co.best_multiexp(VESTA, sc, bases)) == PINNED_FIXED[6]


def _all_pinned_points():
    pts = []
    for vk in VKS:
        pts += [(int(p["x"], 16), int(p["y"], 16)) for p in vk["points"]]
    return pts


def test_group_law_on_reference_pinned_points_python():
    """[q]P = O, [q-1]P = -P for every pinned Vesta point (group order p) and for Pallas (-1, 2) (group order q)."""
    pts = _all_pinned_points()
    assert len(pts) == 288
    for P_ in pts[::6]:                              # 48 of them in Python; the C and GPU tests take all 288
        assert o.ec_mul(o.P, P_, o.Q) is None
        assert o.ec_mul(o.P - 1, P_, o.Q) == o.ec_neg(P_, o.Q)
    G = (o.P - 1, 2)
    assert o.ec_mul(o.Q, G, o.P) is None and o.ec_mul(o.Q - 1, G, o.P) == o.ec_neg(G, o.P)


def test_group_law_on_reference_pinned_points_c():
    pts = _all_pinned_points()
    n = len(pts)
    sf = co.field_of_curve(VESTA, "scalar")
    bases = co.points_to_mont(VESTA, pts)
    neg1 = co.to_mont(sf, co.ints_to_limbs([o.P - 1]))
    one = co.to_mont(sf, co.ints_to_limbs([1]))
    for i in range(n):
        got = co.best_multiexp(VESTA, neg1, bases[i:i + 1])
        assert co.jac_to_affine_ints(VESTA, got) == o.ec_neg(pts[i], o.Q)
        # [q]P = [q-1]P + P = O through the bucket path (P and -P meet in the running sum)
        both = co.best_multiexp(VESTA, np.concatenate([neg1, one]), np.concatenate([bases[i:i + 1], bases[i:i + 1]]))
        assert co.jac_to_affine_ints(VESTA, both) is None
    # [2^k]P by repeated doubling (Python definition arithmetic) == best_multiexp with the scalar 2^k, all points at once
    for k in (1, 7, 64, 200, 253):
        sc = co.to_mont(sf, co.ints_to_limbs([pow(2, k, o.P)] * n))
        want = None
        for P_ in pts:
            d = P_
            for _ in range(k):
                d = o.ec_add(d, d, o.Q)
            want = o.ec_add(want, d, o.Q)
        assert co.jac_to_affine_ints(VESTA, co.best_multiexp(VESTA, sc, bases)) == want
    # Pallas generator (-1, 2)
    G = (o.P - 1, 2)
    gm = co.points_to_mont(0, [G])
    s0 = co.field_of_curve(0, "scalar")
    assert co.jac_to_affine_ints(0, co.best_multiexp(0, co.to_mont(s0, co.ints_to_limbs([o.Q - 1])), gm)) == o.ec_neg(G, o.P)


# ---------------------------------------------------------------------------------------------------------------------
# 3. The proof the reference stores (halo2_proofs/tests/plonk_api_proof.bin, verified by its own test at
#    tests/plonk_api.rs:447-462) is ACCEPTED by the restated verifier with the key recomputed here, and rejected
#    once anything is perturbed.  This pins the restated transcript, challenges, argument formulas, multi-point opening,
#    inner-product verification and its multiexp — the checker the device provers / verifiers are compared with — to the reference.
def _plonk_api_vk(params5):
    from halo2_amd.plonk import ConstraintSystem                 # the plain record type; nothing of the product runs here
    from oracle import plonk as op
    g, g_lagrange, w, u = params5
    text = open(os.path.join(GOLDEN, "plonk_api_pinned_vk.txt")).read()
    cs = pa.constraint_system(ConstraintSystem)
    fixed, perm = pa.pinned_commitments(g_lagrange, w, o.Q, o.P)
    vk = {"cs": cs, "vk_repr": pa.transcript_repr(text), "domain": o.EvaluationDomain(cs.degree, 5, o.P),
          "fixed_commitments": fixed, "permutation_commitments": perm}
    assert vk["domain"].extended_k == PLONK_API["extended_k"]
    pts = (co.points_to_mont(VESTA, g), co.points_to_mont(VESTA, [w])[0], co.points_to_mont(VESTA, [u])[0])
    return op, vk, pts


def test_reference_stored_proof_verifies(params5):
    op, vk, (g, w, u) = _plonk_api_vk(params5)
    proof = open(os.path.join(GOLDEN, "plonk_api_proof.bin"), "rb").read()
    instances = [[[2]], [[2]]]                                   # two circuit instances, public input 2 (plonk_api.rs:404, 456)
    assert op.verify_proof_many(VESTA, 5, g, w, u, vk, instances, proof) is True
    assert op.verify_proof_many(VESTA, 5, g, w, u, vk, [[[2]], [[3]]], proof) is False
    assert op.verify_proof_many(VESTA, 5, g, w, u, dict(vk, vk_repr=vk["vk_repr"] + 1), instances, proof) is False
    for pos in (0, 1000, 2500, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert op.verify_proof_many(VESTA, 5, g, w, u, vk, instances, bytes(bad)) is False


def test_restated_prover_on_the_reference_circuit_and_witness(params5):
    """The sequential restatement of plonk::create_proof, on the reference's circuit + witness (two instances, as its test
    proves them): the proof has the stored proof's length (CircuitCost::proof_size(2), plonk_api.rs:492-497) and is accepted by
    the verifier that accepts the stored proof; keygen through the C restatement gives the pinned key."""
    from oracle import ipa
    op, vk, (g, w, u) = _plonk_api_vk(params5)
    fixed, mapping = pa.keygen_columns(o.P)
    sf = co.field_of_curve(VESTA, "scalar")
    ctr = [5]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    adv, inst = pa.witness(o.P)
    ot = ipa.Transcript(VESTA)
    op.create_proof_many(VESTA, 5, g, w, u, vk["cs"], fixed, mapping, vk["vk_repr"], [(adv, inst), (adv, inst)], rng, ot)
    proof = bytes(ot.out)
    assert len(proof) == len(open(os.path.join(GOLDEN, "plonk_api_proof.bin"), "rb").read()) == 4160
    assert op.verify_proof_many(VESTA, 5, g, w, u, vk, [[[2]], [[2]]], proof)
    kv = op.keygen_vk(VESTA, 5, g, w, vk["cs"], fixed, mapping, vk["vk_repr"])
    assert kv["fixed_commitments"] + kv["permutation_commitments"] == PINNED


def test_small_multiexp_restated_on_the_reference_bench_shape(params5):
    """benches/arithmetic.rs:15-33: Params::new(5), g split in halves, small_multiexp(&[c1, c2], &[g_lo[i], g_hi[i]]) for the 16
    pairs.  The double-and-add restatement equals the definition on the hash-to-curve generators."""
    g = params5[0]
    rng = o.SplitMix64(0x736D616C6C)
    c1, c2 = rng.next_field(o.P) if hasattr(rng, "next_field") else rng.next() * rng.next() * rng.next() * rng.next() % o.P, \
        rng.next() * rng.next() * rng.next() * rng.next() % o.P
    for lo, hi in zip(g[:16], g[16:]):
        assert o.small_multiexp([c1, c2], [lo, hi], o.Q) == o.msm_naive([c1, c2], [lo, hi], o.Q)
    assert o.small_multiexp([], [], o.Q) is None
    assert o.small_multiexp([0, o.P - 1], [g[0], g[1]], o.Q) == o.ec_neg(g[1], o.Q)
