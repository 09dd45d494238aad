#!/usr/bin/env python3
"""Times the device permutation argument's commit (halo2_amd/permutation.py = plonk/permutation/prover.rs:46-197) at k = 20:
3 columns, cs_degree 4 (two grand products), random columns and sigma (timing only -- the values need not satisfy a circuit)."""
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
    from halo2_amd.evaluator import EXTENDED, new_evaluator
    from halo2_amd.permutation import Argument, ProvingKey
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only
    k, curve, cs_degree, bf, n_cols = int(os.environ.get("K", "20")), 1, 4, 5, 3
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    params = h.Params(curve, k, g, g, g[1], g[2])
    dom = h.EvaluationDomain(cs_degree, k, sf)
    up = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
    cols = [up(co.random_field(sf, 10 + i, n)) for i in range(n_cols)]
    sig = [up(co.random_field(sf, 20 + i, n)) for i in range(n_cols)]
    pkey = ProvingKey(sig, [], [])
    pool = co.random_field(sf, 30, 64)

    def rng(count):
        return pool[:count]
    res = {"k": k, "columns": n_cols, "cs_degree": cs_degree}
    for rep in range(3):
        ev = new_evaluator(EXTENDED)
        tr = Blake2bWrite(curve)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Argument(n_cols).commit(params, dom, cs_degree, bf, pkey, cols, 12345, 67890, ev, rng, tr)
        torch.cuda.synchronize()
        res[f"commit_s_run{rep}"] = round(time.perf_counter() - t0, 5)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
