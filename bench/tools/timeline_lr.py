"""Stage timeline (H2_TIMELINE) of the two half-empty registered commits one round of the opening argument issues
(halo2_amd/opening.py, "original" schedule): where the 2.2 ms go."""
import os, sys, time, ctypes as C
os.environ["H2_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from halo2_amd.arithmetic import ipa_round_scalars
from oracle import c_oracle as co
lib = C.CDLL(h.LIB_PATH); h.lib().h2_init(0)
lib.h2_debug_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint]
k, curve = 20, 1
n = 1 << k
sf = fields.CURVE_FIELDS[curve][1]
dev = torch.device("cuda:0")
g = co.generate_bases(curve, 1, n)
w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
params = h.Params(curve, k, g, g, w, u)
d_cl = torch.zeros((n + 1, 4), dtype=torch.int64, device=dev)
d_cr = torch.zeros((n + 1, 4), dtype=torch.int64, device=dev)
ch = [co.random_field(sf, 10 + r, 1)[0] for r in range(k)]
blinds = co.random_field(sf, 9, 2)
j = 3
d_p = torch.from_numpy(co.random_field(sf, 40 + j, 1 << (k - j)).view(np.int64)).to(dev)
ipa_round_scalars(d_p, k, j, ch[:j], sf, d_cl, d_cr)
for _ in range(3): params.opening_columns_commit([d_cl, d_cr], [blinds[0], blinds[1]], affine=False).cpu()
buf = (C.c_ulonglong * (2 * 4096))()
lib.h2_debug_timeline(buf, 4096)
t_host0 = time.perf_counter()
out = params.opening_columns_commit([d_cl, d_cr], [blinds[0], blinds[1]], affine=False)
t_host1 = time.perf_counter()
out.cpu()
t_host2 = time.perf_counter()
cnt = lib.h2_debug_timeline(buf, 4096)
ev = sorted((buf[2 * i], buf[2 * i + 1]) for i in range(cnt))
t0 = ev[0][0]
sid = {}
names = {1: "sort>", 2: "acc >", 3: "tail>", 4: "done "}
for t, tag in ev:
    s = sid.setdefault(tag >> 8, len(sid))
    print(f"{(t - t0) / 100.0:9.1f} us  stream {s}  {names[tag & 0xFF]}")
print(f"host: issue {1e3 * (t_host1 - t_host0):.3f} ms, issue + wait + D2H {1e3 * (t_host2 - t_host0):.3f} ms")
