#!/bin/bash
# tools/r6_split_ab.sh -- bfa_align_heads: fronts of all heads first (default) against head by head (BFA_HEADS_SEQ=1), one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in 1 2 3; do for v in 1 0; do
  export BFA_HEADS_SEQ=$v
  for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 0 2>/dev/null | last | msof "heads_seq=$v c5proxy peak $p"; done
  python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "heads_seq=$v c5proxy 3 in flight"
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last | msof "heads_seq=$v realtext inflight1"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | msof "heads_seq=$v realtext 3 in flight"
  BFA_BS=1,16,64 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('heads_seq=$v B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'])"
done; done
