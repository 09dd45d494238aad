#!/bin/bash
# Round 6 (re-run of the round-5 script on the final tree): where the GPU idles inside create_proof (simple-example, k = 20): kernel + memory-copy trace of bench/tools/create_proof_trace.py, the gaps of the last
# proofs (the warm one and the one from host advice columns).  No --pmc beside the trace domains.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_cp_gaps; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench/tools/create_proof_trace.py > $O/run.txt 2>&1
tail -2 $O/run.txt
python $R/bench/tools/trace_gaps.py $O 75 25
