#!/bin/bash
# One box, one call: `gpurun --timeout 540 -- bash bench/tools/final_r04.sh`.  Validation of the tree as it stands (the driver's own
# sequence: pytest -m gpu, smoke, the bench command) and -- last, so that running out of time only costs this -- the rocprofv3 kernel
# statistics of the bench command (profiles/r04_final_*).  The run recorded in profiles/r04_final_validation.txt also carried the
# same-box A/B of the NTT's RAW9 intermediate form (commit a10ecd0, profiles/r04_ab_ntt_raw9.txt), since dropped.
# Everything lands under gpurun_out/final/; each step logs its elapsed time to steps.log as it ends.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/steps.log; }
el start
timeout 60 rocm-smi --showpower --showclocks > $O/smi_idle.txt 2>&1

timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
el "pytest -m gpu rc=$? ($(tail -1 $O/pytest.log))"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
el "smoke rc=$?"
timeout 180 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
el "bench rc=$?"

cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-create-proof > $O/stats_bench.json 2>/dev/null
el "rocprofv3 3 streams rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o bench1 -- python $R/bench.py --streams 1 --steps 40 --no-cpu-baseline --no-create-proof > $O/stats1_bench.json 2>/dev/null
el "rocprofv3 1 stream rc=$?"
find $O -name "*kernel_stats.csv" | head
# keep what travels back small: the traces are tens of MB
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O | tee -a $O/steps.log
