#!/bin/bash
# After `gpurun -- bash bench/tools/final_r06.sh`: copy what is to be judged from gpurun_out/r06_final (scratch) into profiles/ (tracked).
R=$(cd "$(dirname "$0")/../.." && pwd); O=$R/gpurun_out/r06_final; P=$R/profiles
grep '^{' $O/bench.json | tail -1 > $P/r06_final_bench.json
grep '^{' $O/stats_bench.json | tail -1 > $P/r06_final_bench_under_rocprof_3streams.json
grep '^{' $O/stats1_bench.json | tail -1 > $P/r06_final_bench_under_rocprof_1stream.json
cp $O/stats/bench_kernel_stats.csv $P/r06_final_kernel_stats_3streams.csv
cp $O/stats1/bench1_kernel_stats.csv $P/r06_final_kernel_stats_1stream.csv
cp $O/stats_accumulate_union.json $P/r06_final_accumulate_union_3streams.json
cp $O/stats1_accumulate_union.json $P/r06_final_accumulate_union_1stream.json
cp $O/bench_8rank_rehearsal.json $P/r06_bench_8rank_rehearsal_one_gpu.json
{
  echo "# bench/tools/final_r06.sh on one MI355X box (one gpurun call): the driver's own sequence on the final tree of round 6, then native legs, the opening argument and the 8-rank rehearsal"
  cat $O/steps.log
  echo "--- pytest -m gpu"; grep -E "passed|failed" $O/pytest.log | tail -1
  echo "--- smoke"; tail -1 $O/smoke.log
  echo "--- build/h2bench (native, every figure parity-checked against the C oracle)"
  cat $O/h2bench_commit.txt $O/h2bench_ntt.txt $O/h2bench_msm.txt $O/h2bench_msm22.txt $O/h2bench_host.txt | grep -v "amdgpu.ids"
  echo "--- the opening argument at k = 20: bench/tools/opening_probe.py (p_poly resident, h2_open_device; stamps per round), then build/host_mirror_check opening-time 20 4 (host Vecs, h2_open)"
  cat $O/opening_k20.json
  grep -v "amdgpu.ids" $O/opening_host_mirror.txt
} > $P/r06_final_validation.txt
wc -l $P/r06_final_validation.txt
