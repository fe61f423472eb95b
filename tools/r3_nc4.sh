ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')))"; }
for n in 64 100000 64 100000; do export BFA_NC4_MAX_BATCH=$n
python bench.py --ragged --steps 30 2>/dev/null | j "ragged nc4max=$n"
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 128 2>/dev/null | j "c4 shard nc4max=$n"
done
for n in 64 100000; do export BFA_NC4_MAX_BATCH=$n
python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 256 2>/dev/null | j "c4 full nc4max=$n"
done
