import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.VESTA; sf = co.field_of_curve(curve, "scalar")
N = 1 << 16
bases = co.generate_bases(curve, 1, N); sc = co.random_field(sf, 2, N)
d_b = torch.from_numpy(bases.view(np.int64)).cuda(); d_s = torch.from_numpy(sc.view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in (8, 10, 12, 14, 16):
    n = 1 << k
    row = []
    for c in range(4, 15):
        os.environ["H2_MSM_C"] = str(c)
        for i in range(3): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), n, 1, 0, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); R = 8
        for i in range(R): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), n, 1, 0, d_out.data_ptr(), st)
        torch.cuda.synchronize(); row.append(f"c{c}:{(time.perf_counter() - t0) / R * 1e3:.2f}")
    del os.environ["H2_MSM_C"]
    print(f"n=2^{k} default c={h.msm_window_bits(n)}:", " ".join(row), flush=True)
