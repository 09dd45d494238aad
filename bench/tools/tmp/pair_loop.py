import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import sys, time, numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
curve = 1
sf = fields.CURVE_FIELDS[curve][1]
k = int(sys.argv[1])
n = 1 << k
g = co.generate_bases(curve, 1, n)
w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
params = h.Params(curve, k, g, g, w, u)
d = torch.from_numpy(co.random_field(sf, 4, n + 4).view(np.int64)).cuda()
for _ in range(50):
    params.opening_pair_commit(d, k - 3).cpu()
params.close()
