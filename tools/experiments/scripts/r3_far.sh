#!/bin/bash
# tools/r3_far.sh -- mixed-length calls after a change to the wide classes: parity tests, then ragged / C4 shard / C4
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')))"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --ragged --steps 30 2>/dev/null | j "ragged"
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 128 2>/dev/null | j "c4 shard"
done
python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 256 2>/dev/null | j "c4 full"
timeout 600 python tests/soak.py 60 21 2>&1 | tail -1
