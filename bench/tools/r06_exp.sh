R=$GRAFT_REPO_ROOT
cd $R
export H2BENCH_LIB=$R/build/ab/libhalo2_mi355x_ab.so
run() { echo "== $*"; env "$@" timeout 100 build/h2bench msm ${L:-20} 0 | grep "generic best\|FAIL" | grep -v "^ok" | sed 's/.*device-resident/   /'; }
run A=1
run H2_MSM_HOST_SPLIT=40,38,22
run H2_MSM_HOST_SPLIT=42,38,20
run H2_MSM_HOST_SPLIT=45,40,15
run H2_MSM_HOST_SPLIT=38,36,26
run H2_MSM_HOST_SPLIT=30,30,40
run H2_MSM_HOST_CHUNKS=4 H2_MSM_HOST_SPLIT=32,30,24,14
run H2_MSM_HOST_CHUNKS=4 H2_MSM_HOST_SPLIT=35,30,23,12
run H2_MSM_HOST_CHUNKS=2 H2_MSM_HOST_SPLIT=65,35
run A=1
