#!/usr/bin/env python3
"""Times the lookup argument's device steps at k = 20: the field-element sort, permute_expression_pair, and the whole
commit_permuted + commit_product of halo2_amd/lookup.py (one-column lookup into a 2^16-row table)."""
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
    from halo2_amd.arithmetic import permute_expression_pair, sort_field
    from halo2_amd.evaluator import EXTENDED, LAGRANGE, Ast, new_evaluator
    from halo2_amd.lookup import Argument
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only
    k, curve, bf = int(os.environ.get("K", "20")), 1, 5
    n = 1 << k
    usable = n - bf - 1
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    rnd = np.random.default_rng(1)
    table_rows = co.random_field(sf, 3, 1 << 16)
    table = table_rows[np.arange(n) % (1 << 16)]
    inputs = table_rows[rnd.integers(0, 1 << 16, n)]
    d_in = torch.from_numpy(np.ascontiguousarray(inputs).view(np.int64)).to(dev)
    d_tb = torch.from_numpy(np.ascontiguousarray(table).view(np.int64)).to(dev)
    res = {"k": k}

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / reps * 1e3, 4)
    rand = torch.from_numpy(co.random_field(sf, 4, n).view(np.int64)).to(dev)
    res["sort_2^k_random_ms"] = timed(lambda: sort_field(rand.clone(), sf))
    res["clone_ms"] = timed(lambda: rand.clone())
    res["permute_expression_pair_ms"] = timed(lambda: permute_expression_pair(d_in, d_tb, usable, sf))
    g = co.generate_bases(curve, 1, n)
    params = h.Params(curve, k, g, g, g[1], g[2])
    dom = h.EvaluationDomain(4, k, sf)
    pool = co.random_field(sf, 30, 64)
    rng = lambda count: pool[:count]

    def whole():
        vals, cosets = new_evaluator(LAGRANGE), new_evaluator(EXTENDED)
        vl = [vals.register_poly(d_in), vals.register_poly(d_tb)]
        cl = vl                                                   # the coset Asts are only built here, not evaluated
        tr = Blake2bWrite(curve)
        arg = Argument([lambda c: Ast.of(c[0])], [lambda c: Ast.of(c[1])])
        perm = arg.commit_permuted(params, dom, bf, vals, cosets, 7, vl, cl, rng, tr)
        perm.commit_product(params, dom, bf, 11, 13, cosets, rng, tr)
    res["commit_permuted+commit_product_ms"] = timed(whole, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
