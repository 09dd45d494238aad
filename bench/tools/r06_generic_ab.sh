#!/bin/bash
# Round 6, review item 1: generic best_multiexp (h2_msm_device, no registered table) in the grouped form (csrc/msm_generic.hip) against
# round 5's slice split (H2_GENERIC_GROUPED=0 in the laboratory build), alternating on one box; every run checked against the C oracle.
# SIZES (default 20 21 22), CURVES (default 0), REPS (default 2).  Then the parity sweep through the shipped library.
cd "$(dirname "$0")/../.."
AB=build/ab/libhalo2_mi355x_ab.so
export H2BENCH_MSM_DEVICE_ONLY=1
for rep in $(seq 1 ${REPS:-2}); do
  for L in ${SIZES:-20 21 22}; do
    for C in ${CURVES:-0}; do
      echo "== 2^$L curve $C: shipped (grouped)"
      H2BENCH_CLOCK=${CLOCK:-} timeout 200 build/h2bench msm $L $C | grep "generic\|FAIL\|independent\|one stream"
      echo "== 2^$L curve $C: laboratory build, H2_GENERIC_GROUPED=0 (round 5's slice split)"
      H2BENCH_LIB=$AB H2_GENERIC_GROUPED=0 timeout 200 build/h2bench msm $L $C | grep "generic\|FAIL\|independent\|one stream"
      [ -n "$GROUPS_AB" ] && for gs in $GROUPS_AB; do
        echo "== 2^$L curve $C: laboratory build, H2_GENERIC_GROUPS=$gs"
        H2BENCH_LIB=$AB H2_GENERIC_GROUPS=$gs timeout 200 build/h2bench msm $L $C | grep "generic\|FAIL\|independent\|one stream"
      done
    done
  done
done
echo "== parity sweep (small and odd sizes through every entry point)"
timeout 300 build/h2bench parity | grep -c "^ok"
timeout 300 build/h2bench parity | tail -1
