# the last experiment script of round 6 as it was run (see bench/tools/README.md): a by-kernel trace of the shipped generic multiexp at 2^20 and 2^22
R=$GRAFT_REPO_ROOT
export H2BENCH_MSM_DEVICE_ONLY=1
cd /tmp; export TMPDIR=/tmp
for L in 20 22; do
rocprofv3 --kernel-trace -d $R/gpurun_out/r06_call_$L -o t -- $R/build/h2bench msm $L > $R/gpurun_out/r06_call_$L.log 2>&1; grep "generic best\|independent" $R/gpurun_out/r06_call_$L.log
done
