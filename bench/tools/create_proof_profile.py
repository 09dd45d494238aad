import os, sys, time, cProfile, pstats, importlib.util
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
spec = importlib.util.spec_from_file_location("simple_example", os.path.join(ROOT, "examples", "simple_example.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
n = 1 << 20
pv = co.generate_bases(h.VESTA, 0x56455354, n + 2)
prm = h.Params.from_generators(h.VESTA, 20, np.ascontiguousarray(pv[:n]), None, pv[n], pv[n + 1])
res = mod.prove_and_verify(prm, quiet=True)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()})
pr = cProfile.Profile(); pr.enable()
res = mod.prove_and_verify(prm, quiet=True)
pr.disable()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()})
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
