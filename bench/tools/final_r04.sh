#!/bin/bash
# One box, one call: `gpurun --timeout 540 -- bash bench/tools/final_r04.sh`.  Validation of the tree as it stands (the driver's own
# sequence: pytest -m gpu, smoke, the bench command), then the same-box A/B of the NTT's RAW9 intermediate form (H2_NTT_RAW9 0 / 1 / 2
# against the library built from the tree before the change, build/ab/lib_old_ntt.so), the NTT users of the suite under H2_NTT_RAW9=2,
# and -- last, so that running out of time only costs this -- the rocprofv3 kernel statistics of the bench command.
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

{
for rep in 1 2; do
  for m in 0 2; do
    echo "== new, H2_NTT_RAW9=$m (rep $rep)"; H2_NTT_RAW9=$m NTT_SIZES=20,22 timeout 90 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
  done
  cp halo2_amd/libhalo2_mi355x.so /tmp/new.so; cp build/ab/lib_old_ntt.so halo2_amd/libhalo2_mi355x.so
  echo "== old (tree before the RAW9 change) (rep $rep)"; NTT_SIZES=20,22 timeout 90 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
  cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
done
echo "== new, H2_NTT_RAW9=1"; H2_NTT_RAW9=1 NTT_SIZES=18,20,22,24 timeout 90 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
echo "== new, H2_NTT_RAW9=2, more sizes"; H2_NTT_RAW9=2 NTT_SIZES=16,18,24 timeout 90 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
echo "== new, H2_NTT_RAW9=0, more sizes"; H2_NTT_RAW9=0 NTT_SIZES=16,18,24 timeout 90 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
for m in 0 2; do
  echo "== batched, H2_NTT_RAW9=$m"; H2_NTT_RAW9=$m NTT_SIZES=20,22 timeout 90 python bench/tools/ntt_batch_time.py 2>&1 | grep "2\^"
done
} > $O/ab_ntt_raw9.txt 2>&1
el "A/B ntt raw9 done"

H2_NTT_RAW9=2 timeout 300 python -m pytest tests/test_gpu_ntt_sizes.py tests/test_gpu_parity.py tests/test_gpu_poly.py tests/test_gpu_plonk.py \
    tests/test_gpu_examples.py tests/test_gpu_vanishing.py tests/test_gpu_permutation.py tests/test_gpu_lookup_argument.py \
    tests/test_gpu_reference_goldens.py tests/test_gpu_opening.py -m gpu -x -q > $O/pytest_raw9_2.log 2>&1
el "NTT users under H2_NTT_RAW9=2 rc=$? ($(tail -1 $O/pytest_raw9_2.log))"

cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-create-proof > $O/stats_bench.json 2>/dev/null
el "rocprofv3 3 streams rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o bench1 -- python $R/bench.py --streams 1 --steps 40 --no-cpu-baseline --no-create-proof > $O/stats1_bench.json 2>/dev/null
el "rocprofv3 1 stream rc=$?"
find $O -name "*kernel_stats.csv" | head
# keep what travels back small: the traces are tens of MB
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O | tee -a $O/steps.log
