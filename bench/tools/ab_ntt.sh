mkdir -p gpurun_out/r04d
for rep in 1 2; do
  echo "== new"; NTT_SIZES=20,22 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
  cp halo2_amd/libhalo2_mi355x.so /tmp/new.so; cp build/ab/lib_old_ntt.so halo2_amd/libhalo2_mi355x.so
  echo "== old"; NTT_SIZES=20,22 python bench/tools/ntt_time.py 2>&1 | grep "2\^"
  cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
done
