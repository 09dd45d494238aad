"""20 lone 2^20 commits (with blind) over a column table of width H2_COLUMN_C: the workload of a kernel trace
(rocprofv3 --kernel-trace ... -- python bench/tools/one_commit.py; summarise with bench/tools/kstats.py <trace.csv> 50)."""
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
w = np.ascontiguousarray(co.generate_bases(curve, 0x77, 1)[0])
hd = C.c_uint64(0); lib.h2_bases_register_ex(curve, _p(bases), n, 1, int(lib.h2_commit_column_window_bits(n)), C.byref(hd))
lib.h2_bases_set_blind_base(hd, _p(w), 1)
d_c = torch.from_numpy(col.view(np.int64)).cuda()
d_bl = torch.from_numpy(col[7:8].copy().view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    lib.h2_commit_device(hd, d_c.data_ptr(), n, None, d_bl.data_ptr(), 1, 0, d_out.data_ptr(), st)
torch.cuda.synchronize()
