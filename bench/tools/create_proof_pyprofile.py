"""cProfile of one warm create_proof of the simple-example circuit at k = 20 (host-side view: where the Python driver spends its time)."""
import os, sys, importlib.util, cProfile, pstats, io
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
import halo2_amd.plonk as plonk
spec = importlib.util.spec_from_file_location("simple_example", os.path.join(ROOT, "examples", "simple_example.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
n = 1 << 20
pv = co.generate_bases(h.VESTA, 0x56455354, n + 2)
prm = h.Params.from_generators(h.VESTA, 20, np.ascontiguousarray(pv[:n]), None, pv[n], pv[n + 1])
calls = {"n": 0}
orig = plonk.create_proof
prof = cProfile.Profile()
def wrapped(*a, **k):
    calls["n"] += 1
    if calls["n"] == 2:                       # the warm proof
        prof.enable(); r = orig(*a, **k); prof.disable(); return r
    return orig(*a, **k)
mod_globals = mod.prove_and_verify.__globals__
plonk.create_proof = wrapped
res = mod.prove_and_verify(prm, quiet=True)
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:5000])
