"""Round 5: randomised soak of the paths that start at 2^19 points -- the generic multiexp's slice split (h2_msm_device) and the host-slice
range pipeline with its captured launch sequences (h2_msm) -- against the C restatement: random sizes in [2^19, 2^20 + 2^17), both curves,
sparsity / repetition patterns (all-equal scalars -> heavy buckets in every slice group, 90 % zeros, 16-bit scalars, all-ones halves),
identity and duplicate bases, both data forms for the host path.  The shape changes from case to case, so graphs are dropped and
recaptured; every third case repeats the previous shape with fresh scalars, so replays are covered too.
    python bench/tools/soak5.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20260925)
h.lib().h2_init(0)
t_end = time.time() + budget
cases = fails = 0
prev = None


def pattern(sf, n, seed):
    col = co.random_field(sf, seed, n)
    kind = int(rng.integers(0, 6))
    if kind == 1: col[rng.random(n) < 0.9] = 0
    elif kind == 2: col[:] = col[0]
    elif kind == 3: col[:, 1:] = 0; col = co.to_mont(sf, col & 0xFFFF)
    elif kind == 4: col[rng.random(n) < 0.5] = col[0]
    elif kind == 5: col = co.to_mont(sf, np.tile(np.array([[0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0, 0]], dtype=np.uint64), (n, 1)))
    return np.ascontiguousarray(col), kind


while time.time() < t_end:
    if prev is not None and cases % 3 == 2:
        curve, n, g = prev                                  # the same shape again: the captured sequences are replayed on new bytes
    else:
        curve = int(rng.integers(0, 2))
        n = int(rng.integers(1 << 19, (1 << 20) + (1 << 17)))
        g = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), n)
        if rng.random() < 0.3: g[rng.integers(0, n)] = 0
        if rng.random() < 0.3: g[1] = g[0]
        prev = (curve, n, g)
    sf, bf = co.field_of_curve(curve, "scalar"), co.field_of_curve(curve, "base")
    col, kind = pattern(sf, n, int(rng.integers(1, 1 << 30)))
    want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, g))
    canonical = rng.random() < 0.3
    if canonical:
        got = h.best_multiexp(co.from_mont(sf, col), np.ascontiguousarray(co.from_mont(bf, g.reshape(2 * n, 4)).reshape(n, 8)), curve, form=h.FORM_CANONICAL)
        got = co.to_mont(bf, np.ascontiguousarray(got, dtype=np.uint64).reshape(-1, 4)).reshape(-1)
    else:
        got = h.best_multiexp(col, g, curve)
    ok_host = co.jac_to_affine_ints(curve, np.ascontiguousarray(got, dtype=np.uint64)) == want
    d = h.best_multiexp(torch.from_numpy(col.view(np.int64)).cuda(), torch.from_numpy(g.view(np.int64)).cuda(), curve)
    ok_dev = co.jac_to_affine_ints(curve, d.cpu().numpy().view(np.uint64)) == want
    cases += 1
    if not (ok_host and ok_dev):
        fails += 1
        print("MISMATCH curve", curve, "n", n, "pattern", kind, "canonical", canonical, "host ok", ok_host, "device ok", ok_dev, flush=True)
print(f"soak5: {cases} cases ({2 * cases} multiexps of 2^19 .. 2^20+ points), {fails} mismatches")
sys.exit(1 if fails else 0)
