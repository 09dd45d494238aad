#!/bin/bash
# The rocprofv3 runs behind profiles/r04_*: `gpurun -- bash bench/tools/collect_profiles_r04.sh`; outputs land under gpurun_out/r04p/ and
# are summarised into profiles/ by the python steps at the bottom (run those where the repo is).  PMC counters are collected in their
# own passes (no trace domains), as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
# 1. kernel durations of the bench command itself (3 streams, the driver's --steps 20 --warmup 5) and of the single-stream variant
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-create-proof > $O/stats_bench.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o bench1 -- python $R/bench.py --streams 1 --steps 40 --no-cpu-baseline --no-create-proof > $O/stats1_bench.json 2>/dev/null
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes over the minimal workload (timed commits + NTT leg) ...
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --minimal --no-cpu-baseline --streams 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --minimal --no-cpu-baseline --streams 1 > /dev/null 2>&1
# ... and the same two counters over kernels that move a KNOWN number of bytes in the NTT's access patterns (bench/ubench_fetch.hip):
# the calibration the guide asks for before a FETCH_SIZE reading is quoted for a pattern other than 16 B / lane streaming
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/cal_f -o f --output-format csv -- $R/build/ubench/ubench_fetch > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $O/cal_w -o w --output-format csv -- $R/build/ubench/ubench_fetch > /dev/null 2>&1
# 3. issue / wait shares of a wave's cycles (SQ counters, one pass): summarise with bench/tools/sq_summary.py
C="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout 300 rocprofv3 --pmc $C -d $O/sq -o m --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --minimal --no-cpu-baseline --streams 1 > /dev/null 2>&1
find $O -name "*.csv" | head -20
# afterwards, in the repo:
#   python bench/pmc_summary.py gpurun_out/r04p/pmc_f/.../f_counter_collection.csv gpurun_out/r04p/pmc_w/.../w_counter_collection.csv \
#          gpurun_out/r04p/cal_f/.../f_counter_collection.csv gpurun_out/r04p/cal_w/.../w_counter_collection.csv > profiles/r04_pmc_traffic.json
#   python bench/tools/trace_union.py gpurun_out/r04p/stats/.../bench_kernel_trace.csv  > profiles/r04_accumulate_union_3streams.json
#   cp .../bench_kernel_stats.csv profiles/r04_kernel_stats_3streams.csv   (and the 1-stream one)
