"""Wall time of h2_msm_device (unregistered bases) at 2^k points, one call at a time, back to back: median over 5 batches of 20."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = co.field_of_curve(curve, "scalar")
for k in [int(x) for x in (sys.argv[1:] or ["20"])]:
    N = 1 << k
    bases = co.generate_bases(curve, 1, N); sc = co.random_field(sf, 2, N)
    d_b = torch.from_numpy(bases.view(np.int64)).cuda(); d_s = torch.from_numpy(sc.view(np.int64)).cuda()
    d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(40): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), N, 1, 0, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(20): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), N, 1, 0, d_out.data_ptr(), st)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    ts.sort()
    print(f"generic 2^{k}: median {ts[2]:.4f} ms, min {ts[0]:.4f} ms")
