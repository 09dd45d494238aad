import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, halo2_amd as h
for k in (12, 16, 20):
    t0 = time.perf_counter(); p = h.Params.new(h.VESTA, k); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(k, round(t1 - t0, 3), "s", flush=True); p.close()
t0 = time.perf_counter(); p = h.Params.new(h.VESTA, 20); torch.cuda.synchronize(); print("again 20", round(time.perf_counter() - t0, 3))
