#!/bin/bash
# Round 6, review item 1: generic best_multiexp (h2_msm_device, no registered table) by kernel at 2^20 / 2^21 / 2^22 --
# one call at a time and as independent calls on three streams, shader clock and power sampled by h2bench, then a rocprofv3
# kernel trace of the same runs: the last call's launches in order, and per-kernel totals.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_generic
mkdir -p $O
export H2BENCH_MSM_DEVICE_ONLY=1
for L in ${SIZES:-20 21 22}; do
  echo "== 2^$L, clock sampled" 
  H2BENCH_CLOCK=1 $R/build/h2bench msm $L 2>&1 | tail -6
  rocprofv3 --kernel-trace -d $O/t$L -o t -- $R/build/h2bench msm $L > $O/t$L.log 2>&1
  CSV=$(ls $O/t$L/*kernel_trace.csv $O/t$L/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "-- per kernel (all launches of the run)"; python3 $R/bench/tools/kstats.py $CSV | head -24
  cp $CSV $O/t${L}_kernel_trace.csv
done
