import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from halo2_amd.arithmetic import best_multiexp_batch, compute_inner_product, fold_scalars, parallel_generator_collapse, powers
from oracle import c_oracle as co
curve, k = 1, 20
n = 1 << k
sf = fields.CURVE_FIELDS[curve][1]
dev = torch.device("cuda:0")
g = co.generate_bases(curve, 1, n)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
d_pp = up(co.random_field(sf, 4, n)); x3 = co.random_field(sf, 5, 1)[0]
d_b = powers(x3, n, sf, device=dev); d_g = up(g)
d_uw = up(co.generate_bases(curve, 2, 2))
u_j = co.random_field(sf, 6, 1)[0]
def sync(): torch.cuda.synchronize()
d_pp0, d_b0, d_g0 = d_pp.clone(), d_b.clone(), d_g.clone()
for attempt in range(2):          # the second walk is the steady state (workspaces exist, code objects loaded)
  d_pp, d_b, d_g = d_pp0.clone(), d_b0.clone(), d_g0.clone()
  tot = {"ip": 0, "msm": 0, "fold": 0, "collapse": 0, "cat": 0}
  rows = []
  for j in range(k):
      half = 1 << (k - j - 1)
      lo_p, hi_p = d_pp[:half], d_pp[half:2 * half]
      sync(); t0 = time.perf_counter()
      v = torch.stack([compute_inner_product(hi_p, d_b[:half], sf), compute_inner_product(lo_p, d_b[half:2 * half], sf)]).cpu()
      sync(); t1 = time.perf_counter()
      tail = up(co.random_field(sf, 7, 2))
      pl = (torch.cat([hi_p, tail]), torch.cat([d_g[:half], d_uw])); pr = (torch.cat([lo_p, tail]), torch.cat([d_g[half:2 * half], d_uw]))
      sync(); t2 = time.perf_counter()
      lr = best_multiexp_batch([pl, pr], curve, affine=True).cpu()
      sync(); t3 = time.perf_counter()
      d_pp = fold_scalars(d_pp[:2 * half], u_j, sf); d_b = fold_scalars(d_b[:2 * half], u_j, sf)
      sync(); t4 = time.perf_counter()
      d_g = parallel_generator_collapse(d_g[:2 * half], u_j, curve)
      sync(); t5 = time.perf_counter()
      rows.append((half, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
      for key, v_ in zip(("ip", "cat", "msm", "fold", "collapse"), rows[-1][1:]): tot[key] += v_
for r in rows: print("half=2^%-2d ip %.3f cat %.3f msm %.3f fold %.3f collapse %.3f ms" % ((r[0].bit_length() - 1,) + tuple(1e3 * x for x in r[1:])))
print({k_: round(1e3 * v, 2) for k_, v in tot.items()}, "sum", round(1e3 * sum(tot.values()), 2))
