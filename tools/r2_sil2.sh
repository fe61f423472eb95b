ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "segment or sil or golden or hotpath or level2" 2>&1 | tail -2
python tests/sil_time.py 2>&1 | tail -1
python tests/sil_time.py 2>&1 | tail -1
bash tools/r2_silprof.sh
