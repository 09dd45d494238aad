# A/B of the accumulate's occupancy on one box: the shipped build (168 VGPRs, two workgroups per CU), the same build sized for three
# (H2_ACC_WAVES=3), and a build compiled for four waves per SIMD (-DH2_ACC9_WAVES=4: 128 VGPRs, 160 B of spills per lane).
mkdir -p gpurun_out/r04f
echo "== shipped (2 waves / SIMD)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
echo "== shipped, H2_ACC_WAVES=3"; H2_ACC_WAVES=3 python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
cp halo2_amd/libhalo2_mi355x.so /tmp/new.so; cp build/ab/lib_acc_w4.so halo2_amd/libhalo2_mi355x.so
echo "== -DH2_ACC9_WAVES=4 (128 VGPRs)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
