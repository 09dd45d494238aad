#!/usr/bin/env python3
"""h2_commit_batch_device at 2^20: `count` column commits (with blinds) in one call, ms per commit (H2_BATCH_PIPE=0/1: fork over
three streams / staged sort-accumulate-fold pipeline)."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = h.lib(); assert lib.h2_init(0) == 0
curve, n = h.PALLAS, 1 << 20
sf = co.field_of_curve(curve, "scalar")
bases = co.generate_bases(curve, 0x48414C4F32, n)
cols = [co.random_field(sf, 1000 + c, n) for c in range(4)]
w = np.ascontiguousarray(co.generate_bases(curve, 0x77, 1)[0]); blinds = co.random_field(sf, 0xB11D, 4)
hd = C.c_uint64(0)
assert lib.h2_bases_register_ex(curve, _p(bases), n, 1, int(lib.h2_commit_column_window_bits(n)), C.byref(hd)) == 0
assert lib.h2_bases_set_blind_base(hd, _p(w), 1) == 0
dev = torch.device("cuda", 0)
d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]; d_bl = torch.from_numpy(blinds.view(np.int64)).to(dev)
res = {"batch_pipe": os.environ.get("H2_BATCH_PIPE", "0")}
for count in (8, 20, 64):
    d_out = torch.zeros((count, 12), dtype=torch.int64, device=dev)
    arr = C.c_void_p * count
    sc = arr(*[d_cols[i % 4].data_ptr() for i in range(count)]); bl = arr(*[d_bl[i % 4].data_ptr() for i in range(count)]); o = arr(*[d_out[i].data_ptr() for i in range(count)])
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert lib.h2_commit_batch_device(hd, sc, count, n, None, bl, 1, 0, o, None) == 0, lib.h2_last_error()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / count * 1e3)
    got = co.jac_to_affine_ints(curve, d_out[1].cpu().numpy().view(np.uint64))
    res[f"{count}_columns_ms_per_commit"] = round(min(ts[1:]), 4)
    res[f"{count}_ok"] = got == co.jac_to_affine_ints(curve, co.commit(curve, bases, w, cols[1], blinds[1]))
print(json.dumps(res))
