#!/bin/bash
# Round 5: h2_msm's range pipeline -- ranges x window width x who enqueues (H2_MSM_HOST_THREAD) at 2^20
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05_host
{
for th in 1 0; do for c in 13 16; do for q in 4 3 2; do
echo "== H2_MSM_HOST_THREAD=$th H2_MSM_C=$c H2_MSM_HOST_CHUNKS=$q"
H2_MSM_HOST_THREAD=$th H2_MSM_C=$c H2_MSM_HOST_CHUNKS=$q timeout 120 build/h2bench msm 20 0 | grep "generic\|FAIL"
done; done; done
} > gpurun_out/r05_host/c_sweep.txt 2>&1
cat gpurun_out/r05_host/c_sweep.txt
