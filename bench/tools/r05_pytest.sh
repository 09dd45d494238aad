#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05_final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r05_final/pytest.log
