#!/bin/bash
# Round 5: after the heavy-bucket launches were cut to 16 workgroup rows -- the degenerate-column / parity / proof-byte tests, the opening argument at k = 20
# (bench/tools/opening_probe.py) and two small-commit timings (profiles/r05_opening_k20.txt).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05_heavy
{
timeout 900 python -m pytest tests/test_gpu_column_tables.py tests/test_gpu_parity.py tests/test_gpu_opening.py -x -q 2>&1 | tail -3
echo "== opening argument k=20 (bench/tools/opening_probe.py)"
K=20 timeout 300 python bench/tools/opening_probe.py 2>&1 | tail -12
echo "== small commits"
timeout 100 build/h2bench batch 13 8 0 | grep -v "^ok\|^library\|^inputs"
timeout 100 build/h2bench commit 14 8 2 1 1 | grep -v "^ok\|^library\|^inputs"
} > gpurun_out/r05_heavy/out.txt 2>&1
cat gpurun_out/r05_heavy/out.txt
