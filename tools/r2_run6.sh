ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 900 python tests/soak.py 250 1212 2>&1 | tail -1
python bench.py --ragged --no-cpu | grep "^{"
python bench.py --config c4 --steps 4 --chunk 16384 --parity-sample 64 2>/dev/null | grep "^{" | cut -c1-330
python bench.py --config c4 --global-batch 4096 --steps 10 --chunk 4096 --parity-sample 64 2>/dev/null | grep "^{" | cut -c1-330
python tests/sil_time.py | tail -1
python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
python bench.py --no-cpu --steps 20 --warmup 5 --classes 17 | python tools/ubench/extract.py /dev/stdin
