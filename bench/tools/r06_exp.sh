R=$GRAFT_REPO_ROOT
cd $R
export H2BENCH_MSM_DEVICE_ONLY=1 H2BENCH_LIB=$R/build/ab/libhalo2_mi355x_ab.so
run() { echo "== $*"; env "$@" timeout 100 build/h2bench msm ${L:-20} 0 | grep "generic best\|FAIL" | grep -v "^ok"; }
run A=1
for m in 1 4 5 2 8 7 15; do run H2_GG_LOWPRIO=$m; done
run A=1
run H2_GENERIC_GROUPED=0
