#!/bin/bash
# Round 6, review item 3: the counter evidence of the tree that is benchmarked.  rocprofv3 --pmc passes -- each counter set in its OWN run,
# no trace domains, as MI355X_MICROARCH.md prescribes -- over the native driver's workloads (no Python on the box: seconds per pass):
#   commit   `h2bench commit 20 3 1 1`   registered 2^20 commits on one stream: msm_accumulate<.., 256>, the two-pass sort, the fold
#   ntt      `h2bench ntt 20,22`         ntt_pass9 at 2^20 (10 + 10 stages) and 2^22 (8 + 8 + 6)
#   generic  `h2bench msm 20`            generic best_multiexp, grouped form: msm_glv_digits, msm_d1_*, msm_accumulate<.., 512>
#   cal      build/ubench/ubench_fetch   kernels that move a KNOWN number of bytes in these access patterns (the FETCH_SIZE calibration)
# Passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY.  Output: gpurun_out/r06_pmc/<pass>_<load>/...csv;
# afterwards, in the repo:  python bench/tools/r06_pmc_summary.py gpurun_out/r06_pmc > profiles/r06_pmc_traffic.json   (bench.py reads the newest such file)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_pmc
rm -rf $O; mkdir -p $O $R/build/ubench
[ -x $R/build/ubench/ubench_fetch ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/bench/ubench_fetch.hip -o $R/build/ubench/ubench_fetch
export H2BENCH_MSM_DEVICE_ONLY=1
pass() {   # name, counters...
  name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" -d $O/${name}_commit -o p --output-format csv -- $R/build/h2bench commit 20 3 1 1 > $O/${name}_commit.log 2>&1
  timeout 200 rocprofv3 --pmc "$@" -d $O/${name}_ntt -o p --output-format csv -- $R/build/h2bench ntt 20,22 > $O/${name}_ntt.log 2>&1
  timeout 200 rocprofv3 --pmc "$@" -d $O/${name}_generic -o p --output-format csv -- $R/build/h2bench msm 20 > $O/${name}_generic.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_cal -o p --output-format csv -- $R/build/ubench/ubench_fetch > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $O/write_cal -o p --output-format csv -- $R/build/ubench/ubench_fetch > /dev/null 2>&1
grep -h "H2BENCH\|FAIL" $O/*.log | sort | uniq -c
find $O -name "*counter_collection.csv" | sed "s#$O/##" | sort
du -sh $O
