#!/bin/bash
# tools/r3_winlimit.sh -- mixed-length calls against the frame limit up to which K1 tries its sliding window (default 1536): the
# generator's utterances cross the -1000 sentinel near frame 1000, so longer window attempts are redone with the full layout
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')))"; }
for w in 0 1200 1000 800 600; do
  python bench.py --ragged --steps 30 --win-frames $w 2>/dev/null | j "ragged win-frames=$w"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 --win-frames $w 2>/dev/null | j "c4 shard win-frames=$w"
done
for w in 0 1000 800; do
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 --win-frames $w 2>/dev/null | j "c4 full win-frames=$w"
done
