#!/bin/bash
# Shader clock and socket power while commits run back to back (is the sustained accumulate rate a clock / power limit?).
# usage (repo root, GPU box): bench/tools/clock_watch.sh OUTDIR
out=${1:-gpurun_out/clock}; mkdir -p "$out"
poll() { for i in $(seq 1 "$1"); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ' '; echo; sleep 0.3; done; }
echo "== idle" > "$out/clock.log"; poll 3 >> "$out/clock.log"
for mode in "1 1" "1 3" "8 1" "8 2"; do
  set -- $mode
  echo "== K=$1 S=$2 (batch_sweep loop)" >> "$out/clock.log"
  H2_BATCH_COLS=8 SWEEP_SECONDS=5 python bench/tools/batch_sweep.py $1 $2 > "$out/sweep_$1_$2.log" 2>&1 &
  pid=$!
  for i in $(seq 1 200); do [ -e /tmp/sweep_started ] && break; sleep 0.2; done      # import + table registration
  poll 10 >> "$out/clock.log"
  wait $pid; rm -f /tmp/sweep_started
  grep -E "sustained|K=" "$out/sweep_$1_$2.log" | head -2 >> "$out/clock.log"
done
