import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = 1; n = 1 << 20
bases = co.generate_bases(curve, 1, n)
cols = [co.random_field(sf, 2 + i, n) for i in range(4)]
hd = C.c_uint64(0); lib.h2_bases_register(curve, _p(bases), n, 1, C.byref(hd))
d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
d_out = torch.zeros((64, 12), dtype=torch.int64, device="cuda")
for frac in (1.0, 0.75, 0.5, 0.25):
    lib.h2_set_option(b"msm_lane_fraction", frac)
    for ns in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        sps = [C.c_void_p(s.cuda_stream) for s in streams]
        for i in range(2 * ns): lib.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
        torch.cuda.synchronize()
        K = 48
        t0 = time.perf_counter()
        for i in range(K): lib.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"frac={frac} streams={ns}: {1e3*(t2-t0)/K:.3f} ms/commit", flush=True)
