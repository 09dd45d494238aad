"""Randomised soak of the multiexp paths against the C restatement: random sizes, sparsity patterns, repeated scalars, blinds,
prefix commits on registered tables of several sizes, and generic multiexps (both curves).  Not part of the test suite (its
time is spent in the CPU oracle); run on the MI355X box:  python bench/tools/soak.py [seconds]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20260924)
lib = h.lib(); lib.h2_init(0)
t_end = time.time() + budget
cases = fails = 0


def pattern(sf, n, seed):
    col = co.random_field(sf, seed, n)
    kind = int(rng.integers(0, 7))
    if kind == 1: col[rng.random(n) < 0.9] = 0
    elif kind == 2: col[:] = col[0]                                   # one repeated scalar: heavy buckets
    elif kind == 3: col[:, 1:] = 0; col = co.to_mont(sf, col & 0xFFFF)
    elif kind == 4: col[rng.random(n) < 0.5] = col[0]
    elif kind == 5: col[rng.integers(0, n, size=max(1, n // 50))] = 0
    elif kind == 6: col = co.to_mont(sf, np.tile(np.array([[0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0, 0]], dtype=np.uint64), (n, 1)))
    return col, kind


while time.time() < t_end:
    curve = int(rng.integers(0, 2))
    sf = co.field_of_curve(curve, "scalar")
    logn = int(rng.integers(0, 18))
    n = int(rng.integers(1 << logn, (2 << logn))) if logn else int(rng.integers(1, 3))
    g = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), n)
    if rng.random() < 0.3:
        g[rng.integers(0, n)] = 0                                      # an identity base
    if rng.random() < 0.3 and n > 4:
        g[1] = g[0]                                                    # a duplicate
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, 1, C.byref(hd)) == 0
    for _ in range(3):
        col, kind = pattern(sf, n, int(rng.integers(1, 1 << 30)))
        used = n if rng.random() < 0.6 else int(rng.integers(1, n + 1))
        out = np.zeros(12, dtype=np.uint64)
        if rng.random() < 0.5:
            w, bl = co.generate_bases(curve, int(rng.integers(1, 1 << 30)), 1)[0], co.random_field(sf, int(rng.integers(1, 1 << 30)), 1)[0]
            assert lib.h2_commit(hd, _p(col), used, _p(w), _p(bl), 1, 0, _p(out)) == 0
            want = co.commit(curve, g[:used], w, col[:used], bl)
        else:
            assert lib.h2_commit(hd, _p(col), used, None, None, 1, 0, _p(out)) == 0
            want = co.best_multiexp(curve, col[:used], g[:used])
        ok = co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want)
        got2 = h.best_multiexp(np.ascontiguousarray(col[:used]), np.ascontiguousarray(g[:used]), curve)
        ok2 = co.jac_to_affine_ints(curve, got2) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, col[:used], g[:used]))
        cases += 2
        if not (ok and ok2):
            fails += 1
            print("MISMATCH curve", curve, "n", n, "used", used, "pattern", kind, "registered ok", ok, "generic ok", ok2, flush=True)
    lib.h2_bases_free(hd)
print(f"soak: {cases} multiexps checked, {fails} mismatches")
sys.exit(1 if fails else 0)
