#!/usr/bin/env python3
"""Randomised soak of the round-2 paths against independent computations of the same values:
  * best_fft at random sizes 2^1 .. 2^21 with random (non-root) omegas, both fields, against the C restatement;
  * the opening argument at k = 13 .. 17 with a random switch point (hybrid_rounds) against the "original" schedule
    (every round two commits over the original generators) for the same randomness: identical proof bytes.

    python bench/tools/soak2.py [seconds]"""
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
    from oracle import c_oracle as co
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(os.environ.get("SEED", "1234")))
    t_end = time.time() + budget
    n_fft = n_open = bad = 0
    params_cache = {}
    while time.time() < t_end:
        # ---- NTT
        for _ in range(6):
            f = int(rs.randint(0, 2))
            L = int(rs.choice([1, 2, 3, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]))
            a = co.random_field(f, int(rs.randint(1, 1 << 30)), 1 << L)
            omega = co.random_field(f, int(rs.randint(1, 1 << 30)), 1)[0]
            d = torch.from_numpy(a.view(np.int64)).cuda()
            h.best_fft(d, omega, L, f)
            torch.cuda.synchronize()
            if not np.array_equal(d.cpu().numpy().view(np.uint64), co.best_fft(f, a, omega, L)):
                bad += 1
                print("NTT MISMATCH", f, L, flush=True)
            n_fft += 1
        # ---- opening argument
        curve = int(rs.randint(0, 2))
        k = int(rs.randint(13, 18))
        key = (curve, k)
        if key not in params_cache:
            if len(params_cache) >= 3:
                params_cache.pop(next(iter(params_cache))).close()
            g = co.generate_bases(curve, 700 + k, 1 << k)
            w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
            params_cache[key] = h.Params(curve, k, g, g, w, u)
        params = params_cache[key]
        sf = fields.CURVE_FIELDS[curve][1]
        seed = int(rs.randint(1, 1 << 30))
        px = co.random_field(sf, seed, 1 << k)
        blind = h.Blind(co.random_field(sf, seed + 1, 1)[0])
        p = params.commit(px, blind, affine=True)
        jmax = min(k - 1, 12)
        hybrid = int(rs.choice([0, 1, 2, int(rs.randint(1, jmax + 1)), jmax])) if k >= 16 else 0
        proofs = []
        for schedule, hy in (("original", None), ("paired", hybrid if k >= 16 else None)):
            ctr = [seed + 7]

            def rng(count):
                ctr[0] += 1
                return co.random_field(sf, ctr[0], count)
            tr = Blake2bWrite(curve)
            tr.write_point(p)
            x = tr.squeeze_challenge_scalar()
            tr.write_scalar(h.eval_polynomial(px, x, sf))
            create_proof(params, rng, tr, px, blind, x, schedule=schedule, hybrid_rounds=hy)
            proofs.append(tr.finalize())
        if proofs[0] != proofs[1]:
            bad += 1
            print("OPENING MISMATCH", curve, k, hybrid, flush=True)
        n_open += 1
    print(f"soak2: {n_fft} transforms, {n_open} opening arguments (k = 13..17, random switch points), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
