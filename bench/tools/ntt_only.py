import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co, pasta as o
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = h.FP
a = co.random_field(f, 3, 1 << L)
d = torch.from_numpy(a.view(np.int64)).cuda()
omega = fields.scalar_limbs(o.omega_for(o.P, L), f, True)
for i in range(10): h.best_fft(d, omega, L, f)
torch.cuda.synchronize()
