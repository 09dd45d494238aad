#!/bin/bash
# Round 5: the collapsed generators' table over the endomorphism (nine rows by doubling + nine through phi; H2_IPA_GLV_TABLE=0 = the plain sixteen rows):
# parity (opening + plonk tests incl. the golden proofs), then the opening argument at k = 20 / 16 / 18 with and without it, alternating.
mkdir -p gpurun_out
{
  python -m pytest tests/test_gpu_opening.py tests/test_gpu_plonk.py -q -x 2>&1 | tail -3
  for rep in 1 2; do
    for g in 1 0; do
      echo "== H2_IPA_GLV_TABLE=$g k = 20"
      H2_IPA_GLV_TABLE=$g TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
    done
  done
  for k in 16 18; do
    for g in 1 0; do
      echo "== H2_IPA_GLV_TABLE=$g k = $k"
      K=$k H2_IPA_GLV_TABLE=$g TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1 | cut -c1-200
    done
  done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_glv_table.txt
cat gpurun_out/r05_glv_table.txt
