#!/usr/bin/env python3
"""One registered commit alone (h2_commit_device with its blind) at 2^K points under the window width of H2_MSM_C: the input to
choose_c's registered branch.  Run once per width:  for c in 9 10 ... 16; do H2_MSM_C=$c K=14 python bench/tools/c_sweep_registered.py; done"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import halo2_amd as h
    from halo2_amd import fields
    from oracle import c_oracle as co          # input generation only
    curve = 1
    sf = fields.CURVE_FIELDS[curve][1]
    out = []
    for k in [int(x) for x in os.environ.get("K", "12,13,14,15,16,17").split(",")]:
        n = 1 << k
        g = co.generate_bases(curve, 1, n)
        w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
        p = h.Params(curve, k, g, g, w, u)
        d = torch.from_numpy(co.random_field(sf, 4, n).view(np.int64)).cuda()
        b = h.Blind(co.random_field(sf, 5, 1)[0])
        for _ in range(10):
            p.commit(d, b)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            p.commit(d, b)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out.append(f"2^{k}: c={h.lib().h2_commit_window_bits(n)} {sorted(ts)[15] * 1e3:.3f}")
        p.close()
    print("H2_MSM_C=" + os.environ.get("H2_MSM_C", "-") + "  " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
