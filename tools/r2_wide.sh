ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 900 python tests/soak.py 300 71 2>&1 | tail -1
python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
python bench.py --config c4 --steps 6 --parity-sample 256 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
python tools/latency_time.py 2>&1 | tail -6
python tests/sil_time.py | tail -1
python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
