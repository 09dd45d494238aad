#!/bin/bash
# Round 5: the paired commit of small 16-bit tables with 8-bit sub-digits (msm.hip: pair_subdigit_launch) -- parity (paired commits against two
# commits, opening proofs against the restated prover, the k = 16 whole-proof fixture), then the k = 20 opening argument with and without it.
mkdir -p gpurun_out
{
  python -m pytest tests/test_gpu_opening.py tests/test_gpu_plonk.py -q -x 2>&1 | tail -4
  for rep in 1 2; do
    echo "== sub-digit form (default)"
    TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
    echo "== H2_PAIR_SUBDIGITS=0 (the general paired commit in every round)"
    H2_PAIR_SUBDIGITS=0 TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  done
  for J in 6 5 6; do
    echo "== sub-digit form, switch after J = $J rounds (HYBRID=$J)"
    HYBRID=$J TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  done
  echo "== J = 6 with the general paired commit"
  HYBRID=6 H2_PAIR_SUBDIGITS=0 TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  for sub in 1 0; do
    echo "== one paired commit alone, 2^13 / 2^14 / 2^15 / 2^16 points, H2_PAIR_SUBDIGITS=$sub (1 = default)"
    if [ $sub = 0 ]; then export H2_PAIR_SUBDIGITS=0; else unset H2_PAIR_SUBDIGITS; fi
    python - <<'PY'
import time, numpy as np, torch
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co          # input generation only
curve = 1
sf = fields.CURVE_FIELDS[curve][1]
for k in (13, 14, 15, 16):
    n = 1 << k
    g = co.generate_bases(curve, 1, n)
    w, u = co.generate_bases(curve, 2, 1)[0], co.generate_bases(curve, 3, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    d = torch.from_numpy(co.random_field(sf, 4, n + 4).view(np.int64)).cuda()
    for _ in range(5):
        params.opening_pair_commit(d, k - 3)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        params.opening_pair_commit(d, k - 3).cpu()
        ts.append(time.perf_counter() - t0)
    print("2^%d + 4 points: %.4f ms per paired commit (median of 40, result read back)" % (k, sorted(ts)[20] * 1e3))
    params.close()
PY
  done
} > gpurun_out/r05_pair_subdigits.txt 2>&1
tail -40 gpurun_out/r05_pair_subdigits.txt
