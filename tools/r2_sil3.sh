ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
python tests/sil_time.py 2>&1 | tail -1
python tests/sil_time.py 2>&1 | tail -1
timeout 900 python tests/soak.py 150 77 2>&1 | tail -3
bash tools/r2_silprof.sh
