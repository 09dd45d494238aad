import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = 1; n = 1 << 20
bases = co.generate_bases(curve, 1, n)
col = co.random_field(sf, 2, n)
hd = C.c_uint64(0); lib.h2_bases_register(curve, _p(bases), n, 1, C.byref(hd))
d_c = torch.from_numpy(col.view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(12): lib.h2_commit_device(hd, d_c.data_ptr(), n, None, None, 1, 0, d_out.data_ptr(), st)
torch.cuda.synchronize()
