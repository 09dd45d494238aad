# same-box A/B of the table gathers of msm_accumulate: plain loads (shipped) against loads with the non-temporal hint
# (-DH2_ACC_NT=1, build/ab/lib_acc_nt.so); two rounds each
for rep in 1 2; do
  echo "== plain gathers (shipped)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
  cp halo2_amd/libhalo2_mi355x.so /tmp/new.so; cp build/ab/lib_acc_nt.so halo2_amd/libhalo2_mi355x.so
  echo "== non-temporal gathers"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
  cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
done
