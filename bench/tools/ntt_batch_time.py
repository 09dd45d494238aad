"""Per-transform time of h2_ntt_batch_device (independent column FFTs on internal streams, plan 1) against lone transforms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co, pasta as o
for L in [int(x) for x in os.environ.get('NTT_SIZES', '18,20,22').split(',')]:
    a = co.random_field(h.FP, 3, 1 << L)
    omega = fields.scalar_limbs(o.omega_for(o.P, L), h.FP, True)
    cols = [torch.from_numpy(a.view(np.int64)).cuda() for _ in range(6)]
    for i in range(10): h.best_fft_batch(cols, omega, L, h.FP)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); R = 10
        for i in range(R): h.best_fft_batch(cols, omega, L, h.FP)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (R * len(cols)))
    bf = (1 << (L - 1)) * L
    lone = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); R = 10
        for i in range(R):
            for c in cols: h.best_fft(c, omega, L, h.FP)
        torch.cuda.synchronize(); lone = min(lone, (time.perf_counter() - t0) / (R * len(cols)))
    print(f"batched 2^{L}: {best*1e3:.4f} ms per transform  {bf/best/1e9:.1f} G bf/s   (the same six one after another on one stream: {lone*1e3:.4f} ms)", flush=True)
