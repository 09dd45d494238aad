import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = 1
nmax = 1 << 20
bases = co.generate_bases(curve, 1, nmax); scal = co.random_field(sf, 2, nmax)
d_b = torch.from_numpy(bases.view(np.int64)).cuda(); d_s = torch.from_numpy(scal.view(np.int64)).cuda()
hd = C.c_uint64(0)
from halo2_amd.arithmetic import _p
lib.h2_bases_register(curve, _p(bases), nmax, 1, C.byref(hd))
out = torch.zeros(12, dtype=torch.int64, device="cuda")
for lg in (4, 8, 10, 12, 14, 16, 18, 19, 20):
    n = 1 << lg
    res = []
    for mode in ("generic", "registered-prefix"):
        def call():
            if mode == "generic":
                return lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), n, 1, 0, out.data_ptr(), None)
            return lib.h2_commit_device(hd, d_s.data_ptr(), n, None, None, 1, 0, out.data_ptr(), None)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        reps = 10
        t = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / reps * 1e3)
    print(f"n=2^{lg}: generic {res[0]:.3f} ms (c={lib.h2_msm_window_bits(n)}), registered prefix {res[1]:.3f} ms")
