#!/bin/bash
# Round 5: rocprofv3 kernel + memory-copy trace of build/h2bench msm 20 with four ranges (H2_MSM_HOST_CHUNKS=4): the timeline of one h2_msm call from host slices
# (profiles/r05_h2_msm_host_ranges.txt prints it).  No --pmc beside the trace domains.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05_host/trace
H2_MSM_HOST_CHUNKS=4 timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r05_host/trace -o t -- $R/build/h2bench msm 20 0 > $R/gpurun_out/r05_host/trace/run.txt 2>&1
ls -la $R/gpurun_out/r05_host/trace/*/ 2>/dev/null | head
find $R/gpurun_out/r05_host/trace -name "*.csv" | head
