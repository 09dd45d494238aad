import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.VESTA; sf = co.field_of_curve(curve, "scalar")
for logn in (21, 22):
    n = 1 << logn
    t0 = time.time(); g = co.generate_bases(curve, 77, n); col = co.random_field(sf, 78, n); print("gen", round(time.time() - t0, 1), flush=True)
    hd = C.c_uint64(0)
    t0 = time.time(); assert lib.h2_bases_register(curve, _p(g), n, 1, C.byref(hd)) == 0; print("register", round(time.time() - t0, 2), flush=True)
    out = np.zeros(12, dtype=np.uint64)
    assert lib.h2_commit(hd, _p(col), n, None, None, 1, 0, _p(out)) == 0
    t0 = time.time(); want = co.best_multiexp(curve, col, g); print("cpu", round(time.time() - t0, 2), flush=True)
    print(logn, "registered", co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, want), flush=True)
    out2 = h.best_multiexp(col, g, curve)
    print(logn, "generic", co.jac_to_affine_ints(curve, out2) == co.jac_to_affine_ints(curve, want), flush=True)
    lib.h2_bases_free(hd)
# NTT 2^24
import torch
from halo2_amd import fields
from oracle import pasta as o
L = 24
a = co.random_field(h.FP, 5, 1 << L)
omega = fields.scalar_limbs(o.omega_for(o.P, L), h.FP, True)
t0 = time.time(); want = co.best_fft(h.FP, a, omega, L); print("cpu fft", round(time.time() - t0, 2), flush=True)
got = h.best_fft(a.copy(), omega, L, h.FP)
print("ntt 2^24", np.array_equal(got, want))
