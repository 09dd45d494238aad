import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co, pasta
lib = h.lib(); lib.h2_init(0)
for log_n in (20, 22):
    a = co.random_field(h.FP, 7 + log_n, 1 << log_n)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    omega = fields.scalar_limbs(pasta.omega_for(pasta.P, log_n), h.FP)
    for _ in range(3): h.best_fft(d_a, omega, log_n, h.FP)
    torch.cuda.synchronize()
    reps = 20
    t1 = time.perf_counter()
    for _ in range(reps): h.best_fft(d_a, omega, log_n, h.FP)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / reps
    print(f"MAXR={os.environ.get('H2_NTT_MAXR')} LOGT={os.environ.get('H2_NTT_LOGT')} 2^{log_n}: {dt*1e3:.4f} ms  {(1 << (log_n - 1)) * log_n / dt / 1e9:.1f} Gbf/s")
