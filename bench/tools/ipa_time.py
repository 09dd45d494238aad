import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
curve = h.VESTA; sf = 0
n = 1 << 20
g = co.generate_bases(curve, 1, n); u = co.random_field(sf, 5, 1)[0]
d_g = torch.from_numpy(g.view(np.int64)).cuda()
for half_log in (19, 16, 12):
    d = d_g[: 2 << half_log].clone()
    h.parallel_generator_collapse(d.clone(), u, curve); torch.cuda.synchronize()
    t = time.perf_counter(); h.parallel_generator_collapse(d, u, curve); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"collapse half=2^{half_log}: {dt*1e3:.2f} ms  ({(1<<half_log)/dt/1e6:.1f} M scalar-muls/s)")
# all 20 rounds of a k=20 argument
d = d_g.clone(); torch.cuda.synchronize(); t = time.perf_counter()
while d.shape[0] > 1:
    d = h.parallel_generator_collapse(d, u, curve)
torch.cuda.synchronize(); print(f"all 20 rounds: {(time.perf_counter()-t)*1e3:.2f} ms")
t = time.perf_counter(); ref = co.generator_collapse(curve, g[: 1 << 15], u); dt = time.perf_counter() - t
print(f"oracle (C restatement, all host cores) half=2^14: {dt*1e3:.1f} ms -> {(1<<14)/dt/1e6:.3f} M/s")
