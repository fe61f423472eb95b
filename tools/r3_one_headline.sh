ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'K1 ms', (d.get('roofline') or {}).get('kernel_ms'), d.get('parity'))"; }
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j "as built"
BFA_ONE_MAX_BATCH=8192 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j "one kernel"
python bench.py --steps 20 --warmup 5 --no-cpu --inflight 1 2>/dev/null | j "as built, 1 in flight"
BFA_ONE_MAX_BATCH=8192 python bench.py --steps 20 --warmup 5 --no-cpu --inflight 1 2>/dev/null | j "one kernel, 1 in flight"
done
