"""Randomised soak of the GROUPED generic multiexp (csrc/msm_generic.hip; h2_msm_device from 2^18 + 1 device-resident points) against the C
restatement: random sizes in (2^18, 2^20 + 2^17], sparsity and repetition patterns, identity and duplicate bases, both curves, canonical and
Montgomery inputs, and random stream usage -- one call alone (latency form: three groups on the library's streams) or two / three calls enqueued
back to back on different streams (the later ones find a multiexp in flight and take the throughput form).  Not part of the test suite (its time
is spent in the CPU oracle); run on the MI355X box:  python bench/tools/soak6.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import halo2_amd as h
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20260930)
h.lib().h2_init(0)
t_end = time.time() + budget
cases = fails = 0
streams = [torch.cuda.Stream() for _ in range(3)]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def pattern(sf, n, seed):
    col = co.random_field(sf, seed, n)
    kind = int(rng.integers(0, 7))
    if kind == 1: col[rng.random(n) < 0.9] = 0
    elif kind == 2: col[:] = col[0]                                   # one repeated scalar: every slice's entries in one bucket
    elif kind == 3: col[:, 1:] = 0; col = co.to_mont(sf, col & 0xFFFF)   # below 2^16: most window slices empty
    elif kind == 4: col[rng.random(n) < 0.5] = col[0]
    elif kind == 5: col[rng.integers(0, n, size=max(1, n // 50))] = 0
    elif kind == 6: col = co.to_mont(sf, np.tile(np.array([[0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0, 0]], dtype=np.uint64), (n, 1)))
    return col, kind


while time.time() < t_end:
    curve = int(rng.integers(0, 2))
    sf, bf = co.field_of_curve(curve, "scalar"), co.field_of_curve(curve, "base")
    n = int(rng.integers((1 << 18) + 1, (1 << 20) + (1 << 17)))
    g = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), n)
    if rng.random() < 0.3:
        g[rng.integers(0, n)] = 0
    if rng.random() < 0.3:
        g[1] = g[0]
    d_g = dev(g)
    k_calls = int(rng.integers(1, 4))
    cols = [pattern(sf, n, int(rng.integers(1, 1 << 30))) for _ in range(k_calls)]
    canonical = rng.random() < 0.25
    if canonical:
        d_gc = dev(co.from_mont(bf, g.reshape(2 * n, 4)).reshape(n, 8))
    outs = []
    for i, (col, kind) in enumerate(cols):
        with torch.cuda.stream(streams[i]):
            if canonical:
                outs.append(h.best_multiexp(dev(co.from_mont(sf, col)), d_gc, curve, form=h.FORM_CANONICAL))
            else:
                outs.append(h.best_multiexp(dev(col), d_g, curve))
    torch.cuda.synchronize()
    for (col, kind), out in zip(cols, outs):
        got = out.cpu().numpy().view(np.uint64)
        if canonical:
            got = co.to_mont(bf, got.reshape(-1, 4)).reshape(-1)
        ok = co.jac_to_affine_ints(curve, got) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, g))
        cases += 1
        if not ok:
            fails += 1
            print("MISMATCH curve", curve, "n", n, "pattern", kind, "calls side by side", k_calls, "canonical", canonical, flush=True)
print(f"soak6: {cases} grouped generic multiexps checked, {fails} mismatches")
sys.exit(1 if fails else 0)
