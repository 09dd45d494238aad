import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = co.field_of_curve(curve, "scalar")
N = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
bases = co.generate_bases(curve, 1, N); sc = co.random_field(sf, 2, N)
d_b = torch.from_numpy(bases.view(np.int64)).cuda(); d_s = torch.from_numpy(sc.view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(30): lib.h2_msm_device(curve, d_s.data_ptr(), d_b.data_ptr(), N, 1, 0, d_out.data_ptr(), st)
torch.cuda.synchronize()
