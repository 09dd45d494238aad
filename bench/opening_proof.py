#!/usr/bin/env python3
"""Times the device-resident opening argument (halo2_amd/opening.py = poly/commitment/prover.rs:26-151) and the
polynomial helper kernels at k = 20.  Not part of bench.py's contract line; numbers go to DESIGN.md / BASELINE.md."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--curve", type=int, default=1)
    ap.add_argument("--schedule", default=None, help="original | collapse (default: opening.py's choice)")
    a = ap.parse_args()
    import torch
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.opening import create_proof
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only

    k, curve = a.k, a.curve
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    params = h.Params(curve, k, g, g, w, u)     # g_lagrange is not used by the opening argument
    px = co.random_field(sf, 4, n)
    d_px = torch.from_numpy(px.view(np.int64)).to(dev)
    blind = h.Blind(co.random_field(sf, 5, 1)[0])
    pool = co.random_field(sf, 6, n + 64)
    pos = [0]

    def rng(count):
        if count == n:
            return pool[:n]
        pos[0] = (pos[0] + count) % 32
        return pool[n + pos[0]: n + pos[0] + count]

    res = {"k": k, "curve": curve, "schedule": a.schedule or "default"}
    for rep in range(3):
        tr = Blake2bWrite(curve)
        tr.write_point(params.commit(d_px, blind, affine=True).cpu().numpy().view(np.uint64))
        x = tr.squeeze_challenge_scalar()
        tr.write_scalar(h.eval_polynomial(d_px, x, sf).cpu().numpy().view(np.uint64))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        create_proof(params, rng, tr, d_px, blind, x, schedule=a.schedule)
        torch.cuda.synchronize()
        res[f"create_proof_s_run{rep}"] = round(time.perf_counter() - t0, 4)
    res["proof_bytes"] = len(tr.finalize())

    # helper kernels, device-resident, per-call time
    d_a = d_px.clone()
    d_b = torch.from_numpy(co.random_field(sf, 7, n).view(np.int64)).to(dev)
    x = co.random_field(sf, 8, 1)[0]

    def timed(name, fn, reps=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[name + "_ms"] = round(dt * 1e3, 4)
    timed("eval_polynomial", lambda: h.eval_polynomial(d_a, x, sf))
    timed("inner_product", lambda: h.compute_inner_product(d_a, d_b, sf))
    timed("kate_division", lambda: h.kate_division(d_a, x, sf))
    timed("powers", lambda: h.powers(x, n, sf, device=dev))
    timed("scale_add", lambda: h.scale_add(d_a, x, d_b, sf))
    timed("batch_invert", lambda: h.batch_invert(d_a, sf))
    timed("grand_product", lambda: h.grand_product(d_a, n, x, sf))
    # expression evaluation over the extended domain (poly/evaluator.rs): a gate-shaped tree with 4 leaves at 2^(k+1)
    from halo2_amd.evaluator import EXTENDED, Ast, new_evaluator
    dom = h.EvaluationDomain(3, k, sf)
    ext = [torch.from_numpy(co.random_field(sf, 20 + j, dom.extended_len()).view(np.int64)).to(dev) for j in range(4)]
    ev = new_evaluator(EXTENDED)
    la, lb, lc, lq = (ev.register_poly(t) for t in ext)
    tree = Ast.distribute_powers([(Ast.of(la) * Ast.of(lb.with_rotation(1)) - Ast.of(lc.with_rotation(-1))) * Ast.of(lq),
                                  Ast.of(la) + Ast.linear(3), Ast.one()], 0x1234567)
    timed("evaluate_gate_tree_2^%d" % dom.extended_k, lambda: ev.evaluate(tree, dom))
    del ext
    g_dev = torch.from_numpy(g.view(np.int64)).to(dev)
    timed("generator_collapse_2^19", lambda: h.parallel_generator_collapse(g_dev.clone(), x, curve), reps=5)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
