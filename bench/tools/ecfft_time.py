import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import halo2_amd as h
from oracle import c_oracle as co
curve = h.VESTA
for k in (12, 16, 20):
    g = co.generate_bases(curve, 1, 1 << k)
    h.lagrange_basis(g[:16], curve, 4)
    t = time.perf_counter(); gl = h.lagrange_basis(g, curve, k); dt = time.perf_counter() - t
    print(f"k={k}: lagrange_basis {dt*1e3:.1f} ms ({(k << (k-1))/dt/1e6:.1f} M point-butterflies/s)")
t = time.perf_counter(); co.lagrange_basis(curve, g[:1 << 10], 10); dt = time.perf_counter() - t
print(f"oracle k=10: {dt*1e3:.1f} ms ({(10 << 9)/dt/1e6:.3f} M point-butterflies/s)")
