#!/bin/bash
# Round 5, item 5: h2_msm from host slices as a pipeline of point ranges behind the base upload (csrc/msm.hip, msm_host_chunked):
# H2_MSM_HOST_CHUNKS sweep (1 = the round-4 one-piece path), with and without the captured launch sequences (H2_MSM_HOST_GRAPHS=0).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_host
mkdir -p $O
cd $R
{
for q in 1 2 4 8; do
  echo "== H2_MSM_HOST_CHUNKS=$q"
  H2_MSM_HOST_CHUNKS=$q timeout 120 build/h2bench msm 20 0 | grep -v "^library\|^inputs"
done
echo "== H2_MSM_HOST_CHUNKS=4 H2_MSM_HOST_GRAPHS=0 (plain launches)"
H2_MSM_HOST_CHUNKS=4 H2_MSM_HOST_GRAPHS=0 timeout 120 build/h2bench msm 20 0 | grep -v "^library\|^inputs"
echo "== defaults: 2^19, 2^21 Pallas; 2^20 Vesta"
timeout 120 build/h2bench msm 19 0 | grep -v "^library\|^inputs"
timeout 120 build/h2bench msm 21 0 | grep -v "^library\|^inputs"
timeout 120 build/h2bench msm 20 1 | grep -v "^library\|^inputs"
} > $O/host_msm.txt 2>&1
grep "generic\|==\|FAIL" $O/host_msm.txt
