#!/bin/bash
# Round 5, item 4: generic best_multiexp (device-resident, no registered table) with the slice split of msm_launch:
# H2_GENERIC_SPLIT=0 (round 4's single accumulate) against the lower group's size k = 2..5 at 2^20, then the default at other sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05_generic
{
for k in 0 2 3 4 5 0 3; do
  echo "== H2_GENERIC_SPLIT=$k"
  H2_GENERIC_SPLIT=$k H2_MSM_HOST_CHUNKS=1 timeout 120 build/h2bench msm 20 0 | grep "generic\|FAIL\|^ok: h2_msm_device"
done
for L in 19 21 22; do
  for k in 0 -1; do
    echo "== 2^$L H2_GENERIC_SPLIT=$k (-1 = default)"
    H2_GENERIC_SPLIT=$k H2_MSM_HOST_CHUNKS=1 timeout 160 build/h2bench msm $L 0 | grep "generic\|FAIL\|^ok: h2_msm_device"
  done
done
echo "== Vesta 2^20 default"
H2_MSM_HOST_CHUNKS=1 timeout 120 build/h2bench msm 20 1 | grep "generic\|FAIL\|^ok: h2_msm_device"
echo "== parity sweep (small and odd sizes through every entry point)"
timeout 200 build/h2bench parity | grep -c "^ok"; timeout 200 build/h2bench parity | grep "FAIL\|H2BENCH"
} > gpurun_out/r05_generic/split.txt 2>&1
cat gpurun_out/r05_generic/split.txt
