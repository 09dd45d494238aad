# same-box A/B of the wave priority of the sort / fold kernels (s_setprio in H2_LATENCY_STAGE): shipped 3, build/ab/lib_prio1.so, lib_prio0.so
for rep in 1 2; do
  echo "== prio 3 (shipped)"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"
  cp halo2_amd/libhalo2_mi355x.so /tmp/new.so
  for p in 1 0; do cp build/ab/lib_prio$p.so halo2_amd/libhalo2_mi355x.so; echo "== prio $p"; python bench/tools/batch_sweep.py 1 1,3 2>&1 | grep "K=1"; done
  cp /tmp/new.so halo2_amd/libhalo2_mi355x.so
done
