"""Wall time of one 2^20 commit with blind at a time (median / min over N, after warm-up): the number DESIGN.md quotes as 'lone commit'."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); lib.h2_init(0)
curve = h.PALLAS; sf = 1; k = int(sys.argv[1]) if len(sys.argv) > 1 else 20; n = 1 << k
bases = co.generate_bases(curve, 1, n)
col = co.random_field(sf, 2, n)
w = np.ascontiguousarray(co.generate_bases(curve, 0x77, 1)[0])
hd = C.c_uint64(0); lib.h2_bases_register_ex(curve, _p(bases), n, 1, int(lib.h2_commit_column_window_bits(n)), C.byref(hd))
lib.h2_bases_set_blind_base(hd, _p(w), 1)
d_c = torch.from_numpy(col.view(np.int64)).cuda()
d_bl = torch.from_numpy(col[7:8].copy().view(np.int64)).cuda()
d_out = torch.zeros(12, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def one():
    lib.h2_commit_device(hd, d_c.data_ptr(), n, None, d_bl.data_ptr(), 1, 0, d_out.data_ptr(), st)
for _ in range(200): one()
torch.cuda.synchronize()
ts = []
for _ in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter(); one(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
t0 = time.perf_counter()
for _ in range(100): one()
torch.cuda.synchronize()
print(f"lone commit 2^{k}: median {ts[len(ts)//2]:.4f} ms, min {ts[0]:.4f} ms; back to back on one stream {(time.perf_counter()-t0)*10:.4f} ms")
