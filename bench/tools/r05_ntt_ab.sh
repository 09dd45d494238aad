#!/bin/bash
# Round 5, NTT: parity of the 11- / 12-stage passes (odd stage counts open with a radix-2 round) and a same-box A/B against round 4's
# library (build/ab/lib_r4.so = `SRC_REV=<round-4 head> bench/tools/build_variant.sh r4`), natively (no Python).
#   gpurun --timeout 300 -- bash bench/tools/r05_ntt_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_ntt
mkdir -p $O
cd $R
NEW=halo2_amd/libhalo2_mi355x.so
OLD=build/ab/lib_r4.so
{
echo "== parity: odd and new stage counts, both fields (new library)"
for f in 0 1; do timeout 120 build/h2bench ntt 3,5,7,9,11,13,15,17,19,21,22,23,24 $f 1 | grep -v "^library\|^inputs"; done
echo "== domain transforms through the 11-stage plans"
timeout 60 build/h2bench domain 20 1 0 | grep -v "^library\|^inputs"
timeout 60 build/h2bench domain 20 2 1 | grep -v "^library\|^inputs"
timeout 60 build/h2bench domain 19 2 0 | grep -v "^library\|^inputs"
for rep in 1 2; do
  for L in $OLD $NEW; do
    echo "== $L (rep $rep)"
    H2BENCH_LIB=$R/$L timeout 60 build/h2bench ntt 16,18,20,21,22,23,24 0 0 | grep "ntt 2"
  done
done
for m in 10 11 12; do
  echo "== new library, H2_NTT_MAXR=$m"
  H2_NTT_MAXR=$m timeout 60 build/h2bench ntt 20,21,22,23,24 0 0 | grep "ntt 2"
done
echo "== new library, H2_NTT_MAXR=11 H2_NTT_LOGT_FIRST=0 (one-column first pass)"
H2_NTT_MAXR=11 H2_NTT_LOGT_FIRST=0 timeout 60 build/h2bench ntt 21,22 0 0 | grep "ntt 2"
} > $O/ab.txt 2>&1
tail -5 $O/ab.txt
