"""Three warm create_proofs of the simple-example circuit at k = 20 (no verify after the last), for a kernel trace whose LAST ~39 ms are one
proof:  rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench/tools/create_proof_trace2.py ; trace_window.py <csv> 38"""
import os, sys, importlib.util, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
spec = importlib.util.spec_from_file_location("simple_example", os.path.join(ROOT, "examples", "simple_example.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
n = 1 << 20
pv = co.generate_bases(h.VESTA, 0x56455354, n + 2)
prm = h.Params.from_generators(h.VESTA, 20, np.ascontiguousarray(pv[:n]), None, pv[n], pv[n + 1])
res = mod.prove_and_verify(prm, quiet=True, proofs_only=3) if "proofs_only" in mod.prove_and_verify.__code__.co_varnames else mod.prove_and_verify(prm, quiet=True)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()})
