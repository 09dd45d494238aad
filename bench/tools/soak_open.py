"""Round 5: randomised soak of the whole-argument entry points (h2_open from host vectors, h2_open_device with p_poly resident) against
the sequential restatement of the reference prover (oracle/ipa.py): random k in [1, 13] and both curves for byte parity with the oracle,
k in [14, 17] for agreement between the four routes (one call / step by step, host / resident) where the restatement would take minutes;
polynomials with patterns (zeros, one repeated coefficient, a_i = i), evaluation points 0 / 1 / random, zero and random blinds, the
switch to the collapsed generators forced on or off where the table allows.
    python bench/tools/soak_open.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from halo2_amd.opening import create_proof
from halo2_amd.transcript import Blake2bWrite
from oracle import c_oracle as co
from oracle import ipa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20260926)
h.lib().h2_init(0)
t_end = time.time() + budget
cases = fails = oracle_cases = 0
params_cache = {}


def seeded(sf, seed):
    ctr = [seed]

    def r(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return r


while time.time() < t_end:
    big = rng.random() < 0.25
    k = int(rng.integers(14, 18)) if big else int(rng.integers(1, 14))
    curve = int(rng.integers(0, 2))
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    key = (curve, k)
    if key not in params_cache:
        if len(params_cache) > 6:
            params_cache.pop(next(iter(params_cache))).close()
        g = co.generate_bases(curve, 7000 + 40 * curve + k, n)
        w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
        params_cache[key] = h.Params.from_generators(curve, k, g, None, w, u)
    params = params_cache[key]
    px = co.random_field(sf, int(rng.integers(1, 1 << 30)), n)
    kind = int(rng.integers(0, 4))
    if kind == 1: px[rng.random(n) < 0.8] = 0
    elif kind == 2: px[:] = px[0]
    elif kind == 3: px = fields.to_limbs(range(n), sf, True)
    xk = int(rng.integers(0, 3))
    x = fields.scalar_limbs(xk, sf, True) if xk < 2 else co.random_field(sf, int(rng.integers(1, 1 << 30)), 1)[0]
    blind = h.Blind(np.zeros(4, dtype=np.uint64) if rng.random() < 0.2 else co.random_field(sf, int(rng.integers(1, 1 << 30)), 1)[0])
    seed = int(rng.integers(1, 1 << 30))
    hybrid = None
    if k >= 14 and params.pair_commit_supported() and rng.random() < 0.5:
        hybrid = int(rng.integers(0, min(k - 1, 6)))
    d_px = torch.from_numpy(np.ascontiguousarray(px).view(np.int64)).cuda()
    proofs = []
    for resident, native in ((False, True), (True, True), (bool(rng.integers(0, 2)), False)):
        tr = Blake2bWrite(curve)
        try:
            create_proof(params, seeded(sf, seed), tr, d_px.clone() if resident else px.copy(), blind, x, native=native, hybrid_rounds=hybrid)
            proofs.append(tr.finalize())
        except Exception as e:                      # a column of zeros can make an L_j the identity: every route must refuse alike
            proofs.append(("error", type(e).__name__))
    ok = proofs[0] == proofs[1] == proofs[2]
    if ok and not big and not isinstance(proofs[0], tuple):
        ot = ipa.Transcript(curve)
        ipa.create_proof(curve, k, params.g, params.w, params.u, seeded(sf, seed), ot, px, blind.value, x)
        ok = bytes(ot.out) == proofs[0]
        oracle_cases += 1
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH curve", curve, "k", k, "pattern", kind, "x kind", xk, "hybrid", hybrid, [p if isinstance(p, tuple) else len(p) for p in proofs], flush=True)
for p in params_cache.values():
    p.close()
print(f"soak_open: {cases} arguments x 3 routes ({oracle_cases} of them also against the restated prover), {fails} mismatches")
sys.exit(1 if fails else 0)
