import os, sys, time, ctypes as C
os.environ["H2_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
lib = C.CDLL(h.LIB_PATH); h.lib().h2_init(0)
lib.h2_debug_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint]
L = h.lib()
curve = h.PALLAS; sf = 1; n = 1 << 20
bases = co.generate_bases(curve, 1, n)
cols = [co.random_field(sf, 2 + i, n) for i in range(4)]
hd = C.c_uint64(0); L.h2_bases_register(curve, _p(bases), n, 1, C.byref(hd))
d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
d_out = torch.zeros((64, 12), dtype=torch.int64, device="cuda")
ns = int(sys.argv[1]); frac = float(sys.argv[2])
L.h2_set_option(b"msm_lane_fraction", frac)
streams = [torch.cuda.Stream() for _ in range(ns)]
sps = [C.c_void_p(s.cuda_stream) for s in streams]
for i in range(6): L.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
buf = (C.c_ulonglong * (2 * 4096))()
lib.h2_debug_timeline(buf, 4096)
K = 8
for i in range(K): L.h2_commit_device(hd, d_cols[i % 4].data_ptr(), n, None, None, 1, 0, d_out[i].data_ptr(), sps[i % ns])
cnt = lib.h2_debug_timeline(buf, 4096)
ev = sorted((buf[2 * i], buf[2 * i + 1]) for i in range(cnt))
t0 = ev[0][0]
sid = {}
names = {1: "sort>", 2: "acc >", 3: "tail>", 4: "done "}
for t, tag in ev:
    s = sid.setdefault(tag >> 8, len(sid))
    print(f"{(t - t0) / 100.0:9.1f} us  stream {s}  {names[tag & 0xFF]}")
