#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_pair_prof; mkdir -p $O
for k in 15 14; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$k -o t -- python $R/bench/tools/tmp/pair_loop.py $k > $O/run$k.txt 2>&1
  python $R/bench/tools/kstats_top.py $O/k$k "paired commit 2^$k + 4 points, sub-digit form, 50 calls" $O/top$k.txt | head -22
done
