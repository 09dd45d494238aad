R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_plonk.py -x -q -k "golden" 2>&1 | tail -4
