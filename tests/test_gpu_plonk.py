"""`plonk::create_proof` after synthesis, on the device (halo2_amd/plonk.py), for a circuit shaped like the reference's own
test circuit (halo2_proofs/tests/plonk_api.rs:25-400): a combined add / multiply gate over three advice and four selector
columns, a public-input gate, a lookup of `a` into a fixed table column, copy constraints over a, b, c (two permutation
sets at degree 4).  The proof is read by the oracle's restatement of `plonk::verify_proof` (oracle/plonk.py): accepted for
the true instance; rejected for a wrong instance, a flipped proof bit, and a witness that breaks a gate.
Runs only on a real MI355X (`-m gpu`)."""
import random

import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.plonk import ConstraintSystem, create_proof, keygen_pk
from halo2_amd.transcript import Blake2bWrite
from halo2_amd import verifier as hv
from oracle import c_oracle as co
from oracle import ipa
from oracle import plonk as oplonk

pytestmark = pytest.mark.gpu

from plonk_circuits import A, B, C_, SA, SB, SC, SL, SM, SP, make_cs as _cs, make_witness as _witness  # noqa: F401


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


@pytest.mark.parametrize("k", [5, 7, 11])
def test_create_proof_is_accepted_by_the_restated_verifier(k):
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    cs = _cs()
    usable = n - (cs.blinding_factors + 1)
    rnd = random.Random(k)
    fixed, advice, mapping, instances = _witness(rnd, m, n, usable)
    g = co.generate_bases(curve, 970 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    vk_repr = 0x1234567890ABCDEF ** 3 % m
    pk = keygen_pk(params, cs, fixed, mapping, vk_repr)
    tr = Blake2bWrite(curve)
    create_proof(params, pk, advice, instances, _rng(sf, 7000), tr)
    proof = tr.finalize()
    if k <= 7:
        # byte for byte the proof the sequential restatement of plonk::create_proof writes for the same randomness
        ot = ipa.Transcript(curve)
        oplonk.create_proof(curve, k, g, w, u, cs, fixed, mapping, vk_repr, advice, instances, _rng(sf, 7000), ot)
        assert bytes(ot.out) == proof
    vk = oplonk.keygen_vk(curve, k, g, w, cs, fixed, mapping, vk_repr)
    assert oplonk.verify_proof(curve, k, g, w, u, vk, instances, proof)
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, [[(instances[0][0] + 1) % m]], proof)
    bad = bytearray(proof)
    bad[-33] ^= 1                                                     # inside the opening argument's scalar c
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances, bytes(bad))
    # the product's own verifier (halo2_amd/verifier.py: MSM::eval on the device) agrees on all of it, and its verifying key
    # -- commitments computed on the device -- equals the oracle's
    dvk = hv.keygen_vk(params, pk)
    assert dvk.fixed_commitments == vk["fixed_commitments"] and dvk.permutation_commitments == vk["permutation_commitments"]
    assert hv.verify_proof(params, dvk, instances, proof)
    assert not hv.verify_proof(params, dvk, [[(instances[0][0] + 1) % m]], proof)
    assert not hv.verify_proof(params, dvk, instances, bytes(bad))
    assert not hv.verify_proof(params, dvk, instances, proof[:-1])
    for pos in (5, 40, len(proof) // 2):                              # flips elsewhere: a commitment, an evaluation
        worse = bytearray(proof)
        worse[pos] ^= 4
        assert not hv.verify_proof(params, dvk, instances, bytes(worse))
    # the reference's collapse schedule for the opening argument writes the same proof
    tr_c = Blake2bWrite(curve)
    create_proof(params, pk, advice, instances, _rng(sf, 7000), tr_c, schedule="collapse")
    assert tr_c.finalize() == proof

    # a witness that breaks the arithmetic gate on one row still yields a transcript, which the verifier rejects
    fixed_b, advice_b, mapping_b, instances_b = _witness(random.Random(k), m, n, usable, break_gate=True)
    tr_b = Blake2bWrite(curve)
    create_proof(params, pk, advice_b, instances_b, _rng(sf, 7000), tr_b)
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances_b, tr_b.finalize())
    assert not hv.verify_proof(params, dvk, instances_b, tr_b.finalize())
    params.close()


@pytest.mark.parametrize("variant,k", [("gates_only", 5), ("two_lookups", 6)])
def test_constraint_system_shapes(variant, k):
    """Other shapes of constraint system through the same prover and both verifiers: no lookup / permutation / instance at all;
    two lookups, one of them over products (degree 6: five h pieces, extended domain 2^(k+3))."""
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    cs = _cs(variant)
    usable = n - (cs.blinding_factors + 1)
    fixed, advice, mapping, instances = _witness(random.Random(17 + k), m, n, usable)
    if variant == "gates_only":
        mapping, instances = [], []
    g = co.generate_bases(curve, 990 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    pk = keygen_pk(params, cs, fixed, mapping, 99)
    assert pk.domain.extended_k == k + {3: 1, 6: 3}[cs.degree]
    tr = Blake2bWrite(curve)
    create_proof(params, pk, advice, instances, _rng(sf, 7100), tr)
    proof = tr.finalize()
    vk = oplonk.keygen_vk(curve, k, g, w, cs, fixed, mapping, 99)
    dvk = hv.keygen_vk(params, pk)
    ot = ipa.Transcript(curve)
    oplonk.create_proof(curve, k, g, w, u, cs, fixed, mapping, 99, advice, instances, _rng(sf, 7100), ot)
    assert bytes(ot.out) == proof
    assert oplonk.verify_proof(curve, k, g, w, u, vk, instances, proof)
    assert hv.verify_proof(params, dvk, instances, proof)
    advice[C_][3] = (advice[C_][3] + 1) % m                   # break one row of the arithmetic gate
    tr_b = Blake2bWrite(curve)
    create_proof(params, pk, advice, instances, _rng(sf, 7100), tr_b)
    assert not hv.verify_proof(params, dvk, instances, tr_b.finalize())
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances, tr_b.finalize())
    params.close()


def test_two_circuit_instances_in_one_proof():
    """`create_proof(params, pk, &[circuit_a, circuit_b], &[instances_a, instances_b], ..)` (plonk/prover.rs:35-48): two
    witnesses of the same circuit share one vanishing argument and one opening; swapping the public inputs is rejected."""
    from halo2_amd.plonk import create_proof_many
    curve, k = h.VESTA, 6
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    cs = _cs()
    usable = n - (cs.blinding_factors + 1)
    # same fixed columns / copy constraints need the same selector layout and value classes: reuse the random stream for the
    # layout and perturb only unconstrained witness values
    fixed, advice_a, mapping, inst_a = _witness(random.Random(3), m, n, usable)
    advice_b = [list(col) for col in advice_a]
    free_rows = [r for r in range(usable) if r % 3 != 0 and (r + 1) % 3 != 0 and r % 2 == 0]     # add rows whose b and c are in no cycle
    for r in free_rows:
        advice_b[B][r] = (advice_b[B][r] + 5) % m
        advice_b[C_][r] = (advice_b[A][r] + advice_b[B][r]) % m
    g = co.generate_bases(curve, 999, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    pk = keygen_pk(params, cs, fixed, mapping, 5)
    tr = Blake2bWrite(curve)
    create_proof_many(params, pk, [(advice_a, inst_a), (advice_b, inst_a)], _rng(sf, 7300), tr)
    proof = tr.finalize()
    vk = oplonk.keygen_vk(curve, k, g, w, cs, fixed, mapping, 5)
    dvk = hv.keygen_vk(params, pk)
    ot = ipa.Transcript(curve)
    oplonk.create_proof_many(curve, k, g, w, u, cs, fixed, mapping, 5, [(advice_a, inst_a), (advice_b, inst_a)], _rng(sf, 7300), ot)
    assert bytes(ot.out) == proof
    assert oplonk.verify_proof_many(curve, k, g, w, u, vk, [inst_a, inst_a], proof)
    assert hv.verify_proof_many(params, dvk, [inst_a, inst_a], proof)
    wrong = [[(inst_a[0][0] + 1) % m]]
    assert not hv.verify_proof_many(params, dvk, [inst_a, wrong], proof)
    assert not oplonk.verify_proof_many(curve, k, g, w, u, vk, [wrong, inst_a], proof)
    assert not hv.verify_proof(params, dvk, inst_a, proof)                      # a two-instance proof is not a one-instance proof
    single = Blake2bWrite(curve)
    create_proof(params, pk, advice_a, inst_a, _rng(sf, 7300), single)
    assert len(proof) > len(single.finalize())
    params.close()


@pytest.mark.parametrize("k", [16, 20])
def test_create_proof_bytes_equal_the_restated_prover_golden(k):
    """Whole-proof BYTES at the sizes where the opening argument switches generators (k >= 16) and at BASELINE configs[3]'s k = 20: the
    device prover against the proof the restated prover (oracle/plonk.py, sequential, C-backed) wrote for the same seeded inputs -- minutes to
    tens of minutes of CPU, so it was run once (oracle/make_plonk_proof_fixture.py, committed beside its output tests/golden/plonk_proof_k*.json)
    and the test rebuilds the inputs from the same seeds.  Every commitment, evaluation and opening round of the proof is in those bytes."""
    import hashlib
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plonk_proof_k%d.json" % k)
    if not os.path.exists(path):
        pytest.skip("no golden proof at k = %d (oracle/make_plonk_proof_fixture.py %d)" % (k, k))
    gold = json.load(open(path))
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    cs = _cs()
    usable = n - (cs.blinding_factors + 1)
    fixed, advice, mapping, instances = _witness(random.Random(k), m, n, usable)
    g = co.generate_bases(curve, 970 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    vk_repr = int(gold["vk_repr"], 16)
    assert vk_repr == 0x1234567890ABCDEF ** 3 % m and gold["rng_seed"] == 7000
    pk = keygen_pk(params, cs, fixed, mapping, vk_repr)
    tr = Blake2bWrite(curve)
    create_proof(params, pk, advice, instances, _rng(sf, gold["rng_seed"]), tr)
    proof = tr.finalize()
    assert hashlib.sha256(bytes.fromhex(gold["proof_hex"])).hexdigest() == gold["proof_sha256"]
    assert proof.hex() == gold["proof_hex"]
    dvk = hv.keygen_vk(params, pk)
    assert hv.verify_proof(params, dvk, instances, proof)
    params.close()
