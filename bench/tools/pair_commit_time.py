#!/usr/bin/env python3
"""One commit alone against one paired commit (h2_commit_pair_device, the opening argument's round shape) at 2^k: wall time per
call with a synchronise after each, and the library's own stage timers (sort / accumulate / reduce)."""
import ctypes as C
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
    from halo2_amd._lib import lib
    from oracle import c_oracle as co          # input generation only
    k, curve = int(os.environ.get("K", "20")), 1
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    d = torch.from_numpy(co.random_field(sf, 4, n + 4).view(np.int64)).to(dev)
    blind = h.Blind(co.random_field(sf, 5, 1)[0])
    L = lib()
    res = {"k": k}

    def run(name, fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        L.h2_profile_enable(1)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        r = {"wall_ms_median": round(sorted(ts)[reps // 2] * 1e3, 4)}
        for nm, slot in (("accumulate", 0), ("sort", 2), ("reduce", 3)):
            ms, cnt = C.c_double(0), C.c_uint64(0)
            L.h2_profile_read(slot, C.byref(ms), C.byref(cnt))
            r[nm + "_ms"] = round(ms.value / max(cnt.value, 1), 4)
        L.h2_profile_enable(0)
        res[name] = r
    run("commit", lambda: params.commit(d[:n], blind))
    for sh in (k - 1, k // 2, 0):
        run(f"pair_shift_{sh}", lambda: params.opening_pair_commit(d, sh))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
