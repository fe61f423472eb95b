ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for g in 0 32768 65536 8192 0; do echo "K2 grid $g"; BFA_K2_GRID=$g python tests/sil_time.py 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
