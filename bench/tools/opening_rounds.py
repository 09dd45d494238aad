#!/usr/bin/env python3
"""Per-stage cost of one round of the opening argument in the "original generators" schedule (halo2_amd/opening.py) at k = 20:
the round-scalars kernel, the two half-empty registered commits, the inner products and the folds, each timed alone with a
device synchronise around it."""
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
    from halo2_amd.arithmetic import ipa_round_scalars
    from oracle import c_oracle as co          # input generation only
    k, curve = int(os.environ.get("K", "20")), 1
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    d_cl = torch.zeros((n + 1, 4), dtype=torch.int64, device=dev)
    d_cr = torch.zeros((n + 1, 4), dtype=torch.int64, device=dev)
    ch = [co.random_field(sf, 10 + r, 1)[0] for r in range(k)]
    blinds = co.random_field(sf, 9, 2)
    res = {"k": k}

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / reps * 1e3, 4)
    for j in (0, 1, 5, 10, 15, k - 1):
        d_p = torch.from_numpy(co.random_field(sf, 40 + j, 1 << (k - j)).view(np.int64)).to(dev)
        d_b = torch.from_numpy(co.random_field(sf, 60 + j, 1 << (k - j)).view(np.int64)).to(dev)
        half = 1 << (k - j - 1)
        r = {}
        r["round_scalars_ms"] = timed(lambda: ipa_round_scalars(d_p, k, j, ch[:j], sf, d_cl, d_cr))
        r["commit_lr_ms"] = timed(lambda: params.opening_columns_commit([d_cl, d_cr], [blinds[0], blinds[1]]).cpu())
        r["commit_l_alone_ms"] = timed(lambda: params.opening_columns_commit([d_cl], [blinds[0]]).cpu())
        r["inner_products_ms"] = timed(lambda: torch.stack([h.compute_inner_product(d_p[half:], d_b[:half], sf),
                                                            h.compute_inner_product(d_p[:half], d_b[half:], sf)]).cpu())
        r["folds_ms"] = timed(lambda: (h.fold_scalars(d_p.clone(), ch[j], sf), h.fold_scalars(d_b.clone(), ch[j], sf)))
        res[f"round{j}"] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
