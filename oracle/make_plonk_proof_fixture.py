"""Golden proof bytes of the RESTATED prover (oracle/plonk.py: a sequential, C-backed restatement of plonk::create_proof,
halo2_proofs/src/plonk/prover.rs) for the test circuit of tests/plonk_circuits.py at a size where running it inside a test costs minutes:
    python oracle/make_plonk_proof_fixture.py 16        ->  tests/golden/plonk_proof_k16.json   (~2 min of CPU)
    python oracle/make_plonk_proof_fixture.py 20        ->  tests/golden/plonk_proof_k20.json   (~40 min, several GB)
Everything is derived from seeds (the witness from random.Random(k), generators / blinds from the oracle's seeded generators), so
tests/test_gpu_plonk.py rebuilds the same inputs, runs the DEVICE prover and compares the bytes.  Test infrastructure only."""
import hashlib
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import c_oracle as co          # noqa: E402
from oracle import ipa                     # noqa: E402
from oracle import plonk as oplonk         # noqa: E402
from plonk_circuits import make_cs, make_witness      # noqa: E402

VESTA, FP = 1, 0                           # the circuit runs on Vesta; its scalar field is Fp
P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
VK_REPR = 0x1234567890ABCDEF ** 3 % P
RNG_SEED = 7000


def seeded_rng(seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(FP, ctr[0], count)
    return rng


def inputs(k):
    n = 1 << k
    cs = make_cs()
    usable = n - (cs.blinding_factors + 1)
    fixed, advice, mapping, instances = make_witness(random.Random(k), P, n, usable)
    g = co.generate_bases(VESTA, 970 + k, n)
    w, u = co.generate_bases(VESTA, 60, 1)[0], co.generate_bases(VESTA, 61, 1)[0]
    return cs, fixed, advice, mapping, instances, g, w, u


def main():
    k = int(sys.argv[1])
    cs, fixed, advice, mapping, instances, g, w, u = inputs(k)
    t0 = time.time()
    ot = ipa.Transcript(VESTA)
    oplonk.create_proof(VESTA, k, g, w, u, cs, fixed, mapping, VK_REPR, advice, instances, seeded_rng(RNG_SEED), ot)
    proof = bytes(ot.out)
    out = {"k": k, "curve": "vesta", "circuit": "tests/plonk_circuits.py (make_cs / make_witness with random.Random(k))",
           "generators": "oracle generate_bases(VESTA, 970 + k, n); w, u = seeds 60, 61", "vk_repr": hex(VK_REPR), "rng_seed": RNG_SEED,
           "made_by": "oracle/make_plonk_proof_fixture.py (the restated prover, oracle/plonk.py); seconds: %.0f" % (time.time() - t0),
           "proof_sha256": hashlib.sha256(proof).hexdigest(), "proof_hex": proof.hex()}
    path = os.path.join(ROOT, "tests", "golden", "plonk_proof_k%d.json" % k)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, len(proof), "bytes", out["made_by"])


if __name__ == "__main__":
    main()
