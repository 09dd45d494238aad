#!/usr/bin/env python3
"""Times halo2_amd.plonk.create_proof at k = 20 on the circuit shape of tests/test_gpu_plonk.py (3 advice, 6 fixed, 1 instance
column, 2 gates, 1 lookup, permutation over 3 columns -> 2 sets, degree 4 -> extended domain 2^22).  The witness is random
(the prover does not check satisfaction) except that the lookup input is drawn from its table."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.plonk import ConstraintSystem, create_proof, keygen_pk
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only
    k, curve = int(os.environ.get("K", "20")), 1
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    SA, SB, SC, SM, SP, SL = range(6)
    cs = ConstraintSystem(
        num_fixed_columns=6, num_advice_columns=3, num_instance_columns=1,
        gates=[lambda q: q.advice(0) * q.fixed(SA) + q.advice(1) * q.fixed(SB) + q.advice(0) * q.advice(1) * q.fixed(SM) - q.advice(2) * q.fixed(SC),
               lambda q: q.fixed(SP) * (q.advice(0) - q.instance(0))],
        advice_queries=[(0, 0), (1, 0), (2, 0)], instance_queries=[(0, 0)], fixed_queries=[(c, 0) for c in range(6)],
        permutation_columns=[("advice", 0), ("advice", 1), ("advice", 2)],
        lookups=[([lambda q: q.advice(0)], [lambda q: q.fixed(SL)])], degree=4, blinding_factors=5)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    rnd = np.random.default_rng(5)
    tsize = min(1 << 12, n // 4)                 # every table value must sit in the usable rows
    table_rows = co.random_field(sf, 3, tsize)
    fixed = [up(co.random_field(sf, 10 + i, n)) for i in range(5)] + [up(table_rows[np.arange(n) % tsize])]
    advice = [up(table_rows[rnd.integers(0, tsize, n)]), up(co.random_field(sf, 21, n)), up(co.random_field(sf, 22, n))]
    mapping = np.stack([np.arange(n, dtype=np.int64) + c * n for c in range(3)])
    c2 = min(1000, n // 4)
    mapping[0, :c2], mapping[1, :c2] = mapping[1, :c2].copy(), mapping[0, :c2].copy()      # some 2-cycles
    g = co.generate_bases(curve, 1, n)
    params = h.Params(curve, k, g, g, g[1], g[2])            # timing only: g_lagrange need not be g's Lagrange basis
    t0 = time.perf_counter()
    pk = keygen_pk(params, cs, fixed, mapping, 12345)
    torch.cuda.synchronize()
    res = {"k": k, "extended_k": pk.domain.extended_k, "keygen_pk_s": round(time.perf_counter() - t0, 4)}
    pool = co.random_field(sf, 30, n + 64)
    rng = lambda count: pool[:count]
    for rep in range(3):
        tr = Blake2bWrite(curve)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        create_proof(params, pk, advice, [[7]], rng, tr)
        torch.cuda.synchronize()
        res[f"create_proof_s_run{rep}"] = round(time.perf_counter() - t0, 4)
    proof = tr.finalize()
    res["proof_bytes"] = len(proof)
    from halo2_amd import verifier as hv
    if os.environ.get("REAL_LAGRANGE", "1") == "1":          # verification needs g_lagrange to be g's Lagrange basis
        params2 = h.Params.from_generators(curve, k, g, None, g[1], g[2])
        pk2 = keygen_pk(params2, cs, fixed, mapping, 12345)
        tr = Blake2bWrite(curve)
        create_proof(params2, pk2, advice, [[7]], rng, tr)
        proof = tr.finalize()
        t0 = time.perf_counter()
        vk = hv.keygen_vk(params2, pk2)
        res["keygen_vk_s"] = round(time.perf_counter() - t0, 4)
        for rep in range(2):
            t0 = time.perf_counter()
            ok = hv.verify_proof(params2, vk, [[7]], proof)
            res[f"verify_proof_s_run{rep}"] = round(time.perf_counter() - t0, 4)
        res["verify_accepts_random_witness"] = ok          # a random witness does not satisfy the gates: expected False
    print(json.dumps(res))


if __name__ == "__main__":
    main()
