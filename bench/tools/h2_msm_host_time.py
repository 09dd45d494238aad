"""h2_msm from pageable host pointers (the literal best_multiexp seam) at 2^k points, median of 9 calls; run with H2_MSM_HOST_OVERLAP=0 for the
copy-copy-compute order."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = co.field_of_curve(curve, "scalar")
for k in [int(x) for x in (sys.argv[1:] or ["16", "18", "20"])]:
    n = 1 << k
    bases = co.generate_bases(curve, 3, n); sc = co.random_field(sf, 4, n)
    out = np.zeros(12, dtype=np.uint64)
    for _ in range(3): assert lib.h2_msm(curve, _p(sc), _p(bases), n, 1, 0, _p(out)) == 0
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); assert lib.h2_msm(curve, _p(sc), _p(bases), n, 1, 0, _p(out)) == 0; ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    ok = co.jac_to_affine_ints(curve, out) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, sc, bases))
    print(f"overlap={os.environ.get('H2_MSM_HOST_OVERLAP', '1')} h2_msm 2^{k}: median {ts[4]:.3f} ms (min {ts[0]:.3f}) equals oracle: {ok}", flush=True)
