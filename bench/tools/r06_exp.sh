R=$GRAFT_REPO_ROOT
cd $R
export H2BENCH_MSM_DEVICE_ONLY=1 H2BENCH_LIB=$R/build/ab/libhalo2_mi355x_ab.so
run() { echo "== $*"; env "$@" timeout 100 build/h2bench msm ${L:-20} 0 | grep "generic best\|FAIL" | grep -v "^ok"; }
run A=1
for k in 1 2 3 4 6; do run H2_GG_UNITS=$k; done
run H2_GG_UNITS=4 H2_GG_SPARE=0 H2_GG_LDS=0
run H2_GG_UNITS=2 H2_GG_SPARE=0 H2_GG_LDS=0
run A=1
L=22 run A=1
L=22 run H2_GG_UNITS=2
L=22 run H2_GG_UNITS=4
L=19 run A=1
L=19 run H2_GG_UNITS=2
