cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_optrace; mkdir -p $O
K=20 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench/tools/opening_probe.py > $O/run.txt 2>&1
ls $O
