#!/bin/bash
# Round 5: fe_inv by divsteps (csrc/field_inv.cuh) against the Fermat ladder (build/ab/lib_fermat.so = -DH2_FE_INV_FERMAT=1): device parity
# (build/field_check runs fe_inv against the C oracle), the kernels whose critical path is one lane's inversion, and a whole proof.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05_inv; mkdir -p $O
{
echo "== build/field_check (device field arithmetic, incl. fe_inv, against the C oracle)"
timeout 120 build/field_check 2>&1 | tail -4
echo "== h2bench parity (affine outputs run through xyzz_to_affine)"
timeout 200 build/h2bench parity | grep "FAIL\|H2BENCH"
cat > /tmp/inv_time.py <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import halo2_amd as h
from oracle import c_oracle as co
field, n = h.FP, 1 << 20
a = co.random_field(field, 5, n)
d = torch.from_numpy(a.view(np.int64)).cuda()
for _ in range(3): h.batch_invert(d, field)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): h.batch_invert(d, field)
torch.cuda.synchronize()
print("h2_batch_invert 2^20: %.4f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
got = h.batch_invert(torch.from_numpy(a.view(np.int64)).cuda(), field).cpu().numpy().view(np.uint64)
print("batch_invert == oracle:", bool(np.array_equal(got, co.batch_invert(field, a))))
# one affine commit at 2^12 (latency: the output's normalisation sits at the end of the chain)
curve = h.VESTA
g = co.generate_bases(curve, 7, 1 << 12)
sc = co.random_field(co.field_of_curve(curve, "scalar"), 8, 1 << 12)
ds, dg = torch.from_numpy(sc.view(np.int64)).cuda(), torch.from_numpy(g.view(np.int64)).cuda()
for aff in (False, True):
    for _ in range(3): h.best_multiexp(ds, dg, curve, affine=aff)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): h.best_multiexp(ds, dg, curve, affine=aff)
    torch.cuda.synchronize()
    print("best_multiexp 2^12 affine=%s: %.4f ms" % (aff, (time.perf_counter() - t0) / 20 * 1e3))
PY
cp halo2_amd/libhalo2_mi355x.so /tmp/shipped.so
for arm in shipped fermat; do
  echo "== $arm"
  [ $arm = fermat ] && cp build/ab/lib_fermat.so halo2_amd/libhalo2_mi355x.so
  timeout 200 python /tmp/inv_time.py 2>&1 | tail -5
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --prewarm-ms 50 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
e=d['extra']['create_proof_simple_example_k20']
print('create_proof k=20: %.4f s (from host columns %.4f), verify %.4f, params_from_generators %.3f s' % (e['create_proof_s'], e['create_proof_from_host_columns_s'], e['verify_proof_s'], e['params_from_generators_s']))"
  cp /tmp/shipped.so halo2_amd/libhalo2_mi355x.so
done
} > $O/inv_ab.txt 2>&1
cat $O/inv_ab.txt
