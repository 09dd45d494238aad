R=$GRAFT_REPO_ROOT
cd $R
export H2BENCH_MSM_DEVICE_ONLY=1 H2BENCH_LIB=$R/build/ab/libhalo2_mi355x_ab.so
run() { echo "== $*"; env "$@" timeout 100 build/h2bench msm ${L:-20} 0 | grep "generic\|FAIL\|independent" | grep -v "^ok"; }
for L in 20 22; do
echo "######## 2^$L"
run H2_GENERIC_GROUPED=0
for gs in 6,3 7,2 5,4 4,3,2 5,2,2 9; do
run H2_GENERIC_GROUPS=$gs
run H2_GENERIC_GROUPS=$gs H2_GG_SPARE=8 H2_GG_LDS=1
done
run H2_GENERIC_GROUPED=0
done
