# same-box A/B of the accumulate loop: H2_ACC_LOOP=2 (shipped: the gathered point is consumed before the next gather is issued; cheap identity
# tests) against H2_ACC_LOOP=1 (round 3's loop, build/ab/lib_acc_loop1.so); two rounds each
for rep in 1 2; do
  echo "== loop 2 (shipped)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
  cp halo2_amd/libhalo2_mi355x.so /tmp/new.so; cp build/ab/lib_acc_loop1.so halo2_amd/libhalo2_mi355x.so
  echo "== loop 1 (round 3)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
  cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
done
