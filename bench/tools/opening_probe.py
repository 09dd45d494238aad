#!/usr/bin/env python3
"""Inputs for the opening-argument schedule decision: wall time of every round of the k = 20 argument (stamps taken when the round's
L_j reaches the transcript), and, for a table over 2^m points, the registration time and the time of one registered commit alone."""
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
    from halo2_amd.opening import create_proof
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only
    k, curve = int(os.environ.get("K", "20")), 1
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    res = {"k": k, "form": "one call (h2_open_device_host_s: p_poly resident, s_poly from the host rng)" if os.environ.get("NATIVE", "1") != "0" else "step by step from Python + h2_ipa_rounds_device"}
    for m in ((12, 13, 15, 17) if os.environ.get("TABLES", "1") != "0" else ()):
        if m > k:
            continue
        t0 = time.perf_counter()
        p = h.Params(curve, m, g[:1 << m], g[:1 << m], w, u)
        d = torch.from_numpy(co.random_field(sf, 4, 1 << m).view(np.int64)).to(dev)
        b = h.Blind(co.random_field(sf, 5, 1)[0])
        p.commit(d, b).cpu()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ts = []
        for _ in range(20):
            t2 = time.perf_counter()
            p.commit(d, b).cpu()
            ts.append(time.perf_counter() - t2)
        res[f"table_2^{m}"] = {"register_plus_first_commit_ms": round((t1 - t0) * 1e3, 2), "commit_alone_ms": round(sorted(ts)[10] * 1e3, 4)}
    t0 = time.perf_counter()
    params = h.Params(curve, k, g, g, w, u)
    torch.cuda.synchronize()
    res["params_two_tables_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    px = co.random_field(sf, 4, n)
    d_px = torch.from_numpy(px.view(np.int64)).to(dev)
    blind = h.Blind(co.random_field(sf, 5, 1)[0])
    pool = co.random_field(sf, 6, n + 64)

    def rng(count):
        if count == n:
            return pool[:n]
        return pool[n: n + count]

    class Stamped:                             # a round ends when its L_j reaches the transcript (the blinds are drawn up front now);
        def __init__(self, curve_):            # no `handle` attribute: the round loop then calls back into Python (a few us a round)
            self.t = Blake2bWrite(curve_)
        def write_point(self, point):
            marks.append(time.perf_counter())
            return self.t.write_point(point)
        def write_scalar(self, scalar): return self.t.write_scalar(scalar)
        def squeeze_challenge_scalar(self): return self.t.squeeze_challenge_scalar()
    marks = []
    for rep in range(3):
        marks.clear()
        tr = Stamped(curve)
        x = co.random_field(sf, 8, 1)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        create_proof(params, rng, tr, d_px, blind, x, schedule=os.environ.get("SCHEDULE") or None,
                     hybrid_rounds=int(os.environ["HYBRID"]) if "HYBRID" in os.environ else None, native=os.environ.get("NATIVE", "1") != "0")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    res["total_ms"] = round((t1 - t0) * 1e3, 3)
    # marks: the S commitment, then L_j, R_j per round; round j's L_j arrives when everything of rounds < j and round j's commit is done
    ls = marks[1::2]
    res["until_L0_ms"] = round((ls[0] - t0) * 1e3, 3)
    res["L_to_L_ms"] = [round((b_ - a_) * 1e3, 3) for a_, b_ in zip(ls, ls[1:] + [t1])]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
