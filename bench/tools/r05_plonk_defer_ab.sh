#!/bin/bash
# Round 5: create_proof (simple-example, k = 20) with the evaluations read back together (transcript.DeferredScalars) and one by one (H2_PLONK_DEFER=0), same box, alternating
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for d in 1 0; do
    echo -n "H2_PLONK_DEFER=$d: "
    H2_PLONK_DEFER=$d python - <<'PY'
import os, sys, importlib.util
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
import halo2_amd as h
from oracle import c_oracle as co
spec = importlib.util.spec_from_file_location("simple_example", os.path.join(ROOT, "examples", "simple_example.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
n = 1 << 20
pv = co.generate_bases(h.VESTA, 0x56455354, n + 2)
prm = h.Params.from_generators(h.VESTA, 20, np.ascontiguousarray(pv[:n]), None, pv[n], pv[n + 1])
rs = [mod.prove_and_verify(prm, quiet=True) for _ in range(3)]
print("create_proof_s", [round(r["create_proof_s"], 4) for r in rs], "from host columns", [round(r["create_proof_from_host_columns_s"], 4) for r in rs], "ok", all(r["ok"] for r in rs))
PY
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_plonk_defer_ab.txt
