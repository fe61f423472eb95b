ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 900 python tests/soak.py 300 51 2>&1 | tail -1
for rep in 1 2; do
python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
done
python bench.py --config c4 --steps 6 --parity-sample 256 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
python bench.py --no-cpu --steps 20 --warmup 5 --inflight 1 | python tools/ubench/extract.py /dev/stdin
bash tools/r2_ragprof.sh 2>&1 | tail -8
