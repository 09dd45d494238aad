#!/bin/bash
# Round 5: rocprofv3 kernel + memory-copy trace of the opening argument at k = 20 (bench/tools/opening_probe.py): which kernels a late round is made of
# (profiles/r05_opening_k20.txt quotes it: planes 68 us, accumulate 54, line sums 42, finish 25, sort 52).  No --pmc beside the trace domains.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_optrace; mkdir -p $O
K=20 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench/tools/opening_probe.py > $O/run.txt 2>&1
ls $O
