#!/bin/bash
# `gpurun --timeout 240 -- bash bench/tools/r05_final_opening_kernels.sh`: where the k = 20 opening argument of the final tree spends its time, by kernel --
# rocprofv3 kernel + memory-copy trace of bench/tools/opening_probe.py (three repetitions; the tables below are the last one alone), then the probe untraced.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_open_kernels
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
TABLES=0 timeout 150 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o open -- python $R/bench/tools/opening_probe.py 2>/dev/null | tail -1 > $O/probe_traced.json
cd $R
W=$(python -c "import json; print(json.load(open('$O/probe_traced.json'))['total_ms'] + 0.3)")
{
  echo "# bench/tools/r05_final_opening_kernels.sh, one MI355X: the opening argument at k = 20 (h2_open_device_host_s through the Python mirror), final tree of round 5"
  echo "--- the probe under the tracer (per-round stamps)"; cat $O/probe_traced.json
  echo "--- per kernel, last repetition (window = its wall time + 0.3 ms)"
  python bench/tools/trace_window_stats.py $O/trace $W
  echo "--- busy / idle of the same window"
  python bench/tools/trace_gaps.py $O/trace $W 25
  echo "--- the probe untraced"
  TABLES=0 timeout 100 python bench/tools/opening_probe.py 2>/dev/null | tail -1
} > $O/summary.txt 2>&1
find $O -name "*.csv" -delete
cat $O/summary.txt | cut -c1-200 | head -80
