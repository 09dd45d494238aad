#!/bin/bash
# Round 5: HBM traffic of a 2^22 transform as three passes (8 + 8 + 6, the default) and as two (H2_NTT_MAXR=11: 11 + 11), FETCH_SIZE and
# WRITE_SIZE in separate passes (no trace domains beside --pmc), natively (build/h2bench ntt 22): do two passes move fewer bytes, as the
# round-4 review expected, while taking longer (profiles/r05_ntt_two_pass_ab.txt)?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_nttpmc
mkdir -p $O
for m in 10 11; do
  H2_NTT_MAXR=$m timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/f$m -o f --output-format csv -- $R/build/h2bench ntt 22 0 0 > /dev/null 2>&1
  H2_NTT_MAXR=$m timeout 120 rocprofv3 --pmc WRITE_SIZE -d $O/w$m -o w --output-format csv -- $R/build/h2bench ntt 22 0 0 > /dev/null 2>&1
done
find $O -name "*counter_collection.csv" | head
