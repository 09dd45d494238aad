R=$GRAFT_REPO_ROOT
cd $R
export H2BENCH_MSM_DEVICE_ONLY=1 H2BENCH_LIB=$R/build/ab/libhalo2_mi355x_ab.so
run() { echo "== $*"; env "$@" timeout 100 build/h2bench msm ${L:-20} 0 | grep "generic\|FAIL\|independent" | grep -v "^ok"; }
L=22 run A=1
L=22 run H2_GENERIC_GROUPS=5,4
L=22 run H2_GENERIC_GROUPS=5,2,2
L=21 run A=1
L=21 run H2_GENERIC_GROUPS=5,4
L=21 run H2_GENERIC_GROUPS=5,2,2
