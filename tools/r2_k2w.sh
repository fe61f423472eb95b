ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --ragged --no-cpu --batch 64 --tlo 2990 --thi 3000 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 880 --tlo 2400 --thi 3000 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu | grep "^{" | cut -c1-140
python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d.get('parity_sample'))"
bash tools/r2_solo.sh 2>&1 | grep "k_dp5\|== \|backtrace" | head -20
