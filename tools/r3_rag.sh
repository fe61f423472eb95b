#!/bin/bash
# tools/r3_rag.sh -- mixed-length calls: merged narrow window kernel (BFA_WIN_MERGED) x where the narrower full-layout
# classes queue (BFA_NARROW_LANES: 1 behind the wide chains, 0 on the window stream)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')))"; }
for m in 0 1; do for n in 1; do
  export BFA_WIN_MERGED=$m BFA_NARROW_LANES=$n
  python bench.py --ragged --steps 30 2>/dev/null | j "ragged merged=$m narrow_lanes=$n"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 2>/dev/null | j "c4 shard merged=$m narrow_lanes=$n"
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 2>/dev/null | j "c4 full merged=$m narrow_lanes=$n"
done; done
unset BFA_WIN_MERGED BFA_NARROW_LANES
bash tools/timeline.sh ragged2 1 python $ROOT/bench.py --ragged --steps 3 | grep -v "^W2026"
