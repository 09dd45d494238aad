import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.VESTA; sf = co.field_of_curve(curve, "scalar")
N = 1 << 20
bases = co.generate_bases(curve, 1, N)
sc = co.random_field(sf, 2, N)
d_b = torch.from_numpy(bases.view(np.int64)).cuda(); d_s = torch.from_numpy(sc.view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in (1, 4, 6, 8, 10, 12, 14, 16, 17, 18, 19, 20):
    n = 1 << k
    for i in range(3): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), n, 1, 0, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    lib.h2_profile_enable(1)
    t0 = time.perf_counter(); R = 10
    for i in range(R): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), n, 1, 0, d_out.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    pr = []
    for slot in (2, 0, 3):
        ms, cnt = C.c_double(0), C.c_uint64(0); lib.h2_profile_read(slot, C.byref(ms), C.byref(cnt)); pr.append(ms.value / max(cnt.value, 1))
    lib.h2_profile_enable(0)
    print(f"n=2^{k}: {dt*1e3:.3f} ms  c={h.msm_window_bits(n)} sort {pr[0]:.3f} acc {pr[1]:.3f} reduce {pr[2]:.3f}", flush=True)
