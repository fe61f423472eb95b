ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 900 python tests/soak.py 300 41 2>&1 | tail -1
python bench.py --ragged --no-cpu --batch 64 --tlo 2990 --thi 3000 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 64 --tlo 2390 --thi 2400 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 880 --tlo 2400 --thi 3000 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu | grep "^{" | cut -c1-140
python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d.get('parity_sample'))"
python bench.py --config c4 --steps 6 --parity-sample 256 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
bash tools/r2_ragprof.sh 2>&1 | tail -24
