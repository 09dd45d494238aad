#!/bin/bash
# Round 5, item 2: does the accumulate's half-used gather traffic (2.4 GB per launch) cost shader clock / time under the socket's power limit?
# The shipped library against builds whose accumulate folds every table index into a WINDOW of the table
# (`bench/tools/build_variant.sh acc_MASK -DH2_ACC_GATHER_MASK=MASKu`; same instruction stream, wrong sums by design, so their parity line
# FAILS): 0x3FFF = 1 MiB (resident in every XCD's L2), 0xFFFFF = 64 MiB and 0x3FFFFF = 256 MiB (Infinity Cache sized), 0x7FFFFF = 512 MiB
# (half the table).  Under the bench line's own schedule (3 streams) and with one stream, 1000-commit regions, shader clock / socket power
# of THIS GPU sampled from a thread.
#   gpurun --timeout 300 -- bash bench/tools/r05_acc_power.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_acc
mkdir -p $O
cd $R
{
for rep in 1 2; do
  for L in halo2_amd/libhalo2_mi355x.so build/ab/lib_accl2.so build/ab/lib_acc_0xFFFFF.so build/ab/lib_acc_0x3FFFFF.so build/ab/lib_acc_0x7FFFFF.so; do
    echo "== $L (rep $rep)"
    H2BENCH_LIB=$R/$L timeout 90 build/h2bench commit 20 1000 5 3,1 0 1 2>&1 | grep -v "^library\|^inputs\|^ok\|^h2_bases"
  done
done
echo "== NTT clocks (shipped library)"
H2BENCH_CLOCK=1 timeout 60 build/h2bench ntt 20,22 0 0 | grep -v "^library"
} > $O/acc_power.txt 2>&1
cat $O/acc_power.txt
