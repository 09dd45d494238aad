#!/bin/bash
# Same-box A/B of two builds of the library through the native driver (no Python: seconds of box time):
#   gpurun --timeout 90 -- 'bench/tools/ab_native.sh build/ab/lib_A.so halo2_amd/libhalo2_mi355x.so commit 20 100 > gpurun_out/ab.txt 2>&1'
# Runs `build/h2bench <mode args>` with each library alternately, REPS (default 2) times each, and prints the measurement lines side
# by side; every run also checks its results against the C oracle (a variant that is fast and wrong says FAIL).
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in $(seq 1 ${REPS:-2}); do
  for L in "$A" "$B"; do
    echo "== $L (rep $rep)"
    H2BENCH_LIB=$R/$L timeout ${RUN_TIMEOUT:-40} build/h2bench "$@" 2>&1 | grep -v "^ok\|^library\|^inputs"
  done
done
