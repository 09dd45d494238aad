#!/bin/bash
# One box, one call: `gpurun --timeout 1500 -- bash bench/tools/final_r06.sh`.  Validation of the tree as it stands -- the driver's own
# sequence (pytest -m gpu, smoke, the bench command) -- then the rocprofv3 kernel statistics of the bench command (three streams and one),
# the native driver's summary and the eight-rank rehearsal of the driver's N = 8 launch line on this one GPU (gloo), whose JSON line
# carries every rank's own step time, table build, clock and power (profiles/r06_final_*).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/steps.log; }
: > $O/steps.log
el start
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
el "pytest -m gpu rc=$? ($(grep -E 'passed|failed' $O/pytest.log | tail -1))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
el "smoke rc=$? ($(tail -1 $O/smoke.log))"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
el "bench rc=$?"
timeout 120 build/h2bench commit 20 20,100 5 3 0 3 > $O/h2bench_commit.txt 2>&1
timeout 120 build/h2bench ntt 16,18,20,21,22,24 0 1 > $O/h2bench_ntt.txt 2>&1
H2BENCH_CLOCK=1 timeout 120 build/h2bench msm 20 0 > $O/h2bench_msm.txt 2>&1
H2BENCH_MSM_DEVICE_ONLY=1 H2BENCH_CLOCK=1 timeout 120 build/h2bench msm 22 0 > $O/h2bench_msm22.txt 2>&1
timeout 120 build/h2bench host 20 > $O/h2bench_host.txt 2>&1
el "h2bench legs done"
# the opening argument (commitment::create_proof as one native call): resident from the Python mirror, with per-round stamps; from host Vecs through the C++ mirror
TABLES=0 timeout 200 python bench/tools/opening_probe.py 2>/dev/null | tail -1 > $O/opening_k20.json
timeout 200 build/host_mirror_check opening-time 20 4 > $O/opening_host_mirror.txt 2>&1
el "opening legs done"
PORT=$(python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])
PY
)
H2_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 8 --steps 5 --warmup 1 --no-cpu-baseline --no-create-proof --prewarm-ms 50 > $O/bench_8rank_rehearsal.out 2> $O/bench_8rank_rehearsal.err
el "8-rank rehearsal rc=$?"
grep '^{' $O/bench_8rank_rehearsal.out | tail -1 > $O/bench_8rank_rehearsal.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-create-proof > $O/stats_bench.json 2>/dev/null
el "rocprofv3 3 streams rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o bench1 -- python $R/bench.py --streams 1 --steps 40 --no-cpu-baseline --no-create-proof > $O/stats1_bench.json 2>/dev/null
el "rocprofv3 1 stream rc=$?"
cd $R
for d in stats stats1; do
  f=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python bench/tools/trace_union.py $f > $O/${d}_accumulate_union.json 2>/dev/null
done
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
du -sh $O | tee -a $O/steps.log
cat $O/steps.log
