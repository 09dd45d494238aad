import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co, pasta as o
for L in [int(x) for x in os.environ.get("NTT_SIZES", "16,18,20,22,24").split(",")]:
    a = co.random_field(h.FP, 3, 1 << L)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    omega = fields.scalar_limbs(o.omega_for(o.P, L), h.FP, True)
    for i in range(30): h.best_fft(d, omega, L, h.FP)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); R = 40
        for i in range(R): h.best_fft(d, omega, L, h.FP)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / R)
    bf = (1 << (L - 1)) * L
    print(f"2^{L}: {best*1e3:.4f} ms  {bf/best/1e9:.1f} G bf/s", flush=True)
