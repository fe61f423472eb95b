#!/bin/bash
# tools/r3_prio.sh -- mixed-length calls: wave priority of the narrow classes' DP waves (BFA_PRIO & 3: 0 -> 3, 1 -> 2, 2 -> 1)
# and of the wide classes' producers (BFA_PRIO & 4 -> 1, & 8 -> 2) -- does the widest class's chain finish earlier?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')))"; }
for pm in 0 1 2 5 6 9 10 0; do
  export BFA_PRIO=$pm
  python bench.py --ragged --steps 30 2>/dev/null | j "ragged prio=$pm"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 2>/dev/null | j "c4 shard prio=$pm"
done
for pm in 0 1 2 6; do
  export BFA_PRIO=$pm
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 2>/dev/null | j "c4 full prio=$pm"
done
