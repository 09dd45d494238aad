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
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    sps = [C.c_void_p(s.cuda_stream) for s in streams]
    for prof in (0, 1):
        lib.h2_profile_enable(prof)
        for i in range(6): lib.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
        torch.cuda.synchronize()
        K = 40
        t0 = time.perf_counter()
        for i in range(K): lib.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"streams={ns} prof={prof}: host enqueue {1e3*(t1-t0)/K:.3f} ms/commit, total {1e3*(t2-t0)/K:.3f} ms/commit")
    lib.h2_profile_enable(0)
