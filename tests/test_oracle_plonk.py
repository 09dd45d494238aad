"""CPU-only: the oracle's sequential restatement of `plonk::create_proof` (oracle/plonk.py, composing the restated permutation,
lookup, vanishing, multiopen and opening provers) against the oracle's restatement of `plonk::verify_proof` -- prover and
verifier restated independently from the reference's two files -- on the plonk_api-shaped circuit: accepted; rejected for a
wrong public input, a flipped bit and a witness that breaks a gate; two circuit instances in one proof.  The GPU suite then
demands byte-identical proofs from the device prover."""
import random

import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import ipa
from oracle import plonk as oplonk
from plonk_circuits import C_, make_cs, make_witness


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


def _setup(k, variant="full"):
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n = 1 << k
    cs = make_cs(variant)
    usable = n - (cs.blinding_factors + 1)
    fixed, advice, mapping, instances = make_witness(random.Random(k), m, n, usable)
    if variant == "gates_only":
        mapping, instances = [], []
    g = co.generate_bases(curve, 970 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    return curve, sf, m, cs, fixed, advice, mapping, instances, g, w, u


@pytest.mark.parametrize("k,variant", [(5, "full"), (5, "gates_only"), (6, "two_lookups")])
def test_restated_prover_and_verifier_agree(k, variant):
    curve, sf, m, cs, fixed, advice, mapping, instances, g, w, u = _setup(k, variant)
    t = ipa.Transcript(curve)
    oplonk.create_proof(curve, k, g, w, u, cs, fixed, mapping, 77, advice, instances, _rng(sf, 7000), t)
    proof = bytes(t.out)
    vk = oplonk.keygen_vk(curve, k, g, w, cs, fixed, mapping, 77)
    assert oplonk.verify_proof(curve, k, g, w, u, vk, instances, proof)
    if instances:
        assert not oplonk.verify_proof(curve, k, g, w, u, vk, [[(instances[0][0] + 1) % m]], proof)
    bad = bytearray(proof)
    bad[-33] ^= 1
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances, bytes(bad))
    broken = [list(c) for c in advice]
    broken[C_][4] = (broken[C_][4] + 1) % m
    t2 = ipa.Transcript(curve)
    oplonk.create_proof(curve, k, g, w, u, cs, fixed, mapping, 77, broken, instances, _rng(sf, 7000), t2)
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances, bytes(t2.out))


def test_restated_prover_two_instances():
    k = 5
    curve, sf, m, cs, fixed, advice, mapping, instances, g, w, u = _setup(k)
    t = ipa.Transcript(curve)
    oplonk.create_proof_many(curve, k, g, w, u, cs, fixed, mapping, 3, [(advice, instances), (advice, instances)], _rng(sf, 1), t)
    vk = oplonk.keygen_vk(curve, k, g, w, cs, fixed, mapping, 3)
    assert oplonk.verify_proof_many(curve, k, g, w, u, vk, [instances, instances], bytes(t.out))
    assert not oplonk.verify_proof(curve, k, g, w, u, vk, instances, bytes(t.out))
