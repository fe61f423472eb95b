ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], (d.get('roofline') or {}).get('kernel_ms'), (d.get('roofline') or {}).get('frac'), d.get('status_ok', (d.get('parity_sample') or d.get('parity') or {})))"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2

python tools/latency_device.py 2>&1 | grep "B=\|sync"
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j headline
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j headline
python bench.py --ragged --steps 30 2>/dev/null | j ragged
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 128 2>/dev/null | j "c4 shard"
python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 256 2>/dev/null | j "c4 full"
python bench.py --config realtext --steps 20 --warmup 5 2>/dev/null | j realtext
python tests/sil_time.py 2>/dev/null | grep "^{" | tail -1 | cut -c1-200
