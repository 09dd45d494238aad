import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = co.field_of_curve(curve, "scalar")
for logn in (14, 18, 19, 20):
    n = 1 << logn
    g = co.generate_bases(curve, 55, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register_ex(curve, _p(g), n, 1, int(os.environ.get("WINDOW_BITS", lib.h2_commit_column_window_bits(n))), C.byref(hd)) == 0
    for name in ("dense", "zeros90", "small"):
        col = co.random_field(sf, 600 + logn, n)
        if name == "zeros90": col[np.arange(n) % 10 != 0] = 0
        if name == "small": col[:, 1:] = 0; col = co.to_mont(sf, (col & 0xFFFF))
        out = np.zeros(12, dtype=np.uint64)
        if name == "dense":   # with a blind, and a prefix commit
            w = co.generate_bases(curve, 9, 1)[0]; bl = co.random_field(sf, 10, 1)[0]
            assert lib.h2_commit(hd, _p(col), n, _p(w), _p(bl), 1, 0, _p(out)) == 0
            want = co.commit(curve, g, w, col, bl)
            print(logn, "dense+blind", co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want), flush=True)
            n2 = n // 2 + 5
            assert lib.h2_commit(hd, _p(col), n2, None, None, 1, 0, _p(out)) == 0
            want = co.best_multiexp(curve, col[:n2], g[:n2])
            print(logn, "prefix", co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want), flush=True)
            n3 = 37
            assert lib.h2_commit(hd, _p(col), n3, None, None, 1, 0, _p(out)) == 0
            want = co.best_multiexp(curve, col[:n3], g[:n3])
            print(logn, "tiny prefix", co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want), flush=True)
        assert lib.h2_commit(hd, _p(col), n, None, None, 1, 0, _p(out)) == 0
        want = co.best_multiexp(curve, col, g)
        print(logn, name, "c =", h.msm_window_bits(n), co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want), flush=True)
    lib.h2_bases_free(hd)
