#!/bin/bash
# tools/r6_xlane.sh -- full silence-anchored batches: the wide exact-window kernels on lanes of their own (0) / with k_mix on the
# head's third lane, behind it (1) or ahead of it (2); experiment build, one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in 1 2; do for m in 0 1 2; do
  export BFA_XWIN_LANE_MODE=$m
  for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 16 2>/dev/null | last | msof "xwin_lane_mode=$m c5proxy peak $p"; done
  python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "xwin_lane_mode=$m c5proxy 3 in flight"
done; done
