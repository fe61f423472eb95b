#!/bin/bash
# tools/r6_splitlat.sh -- full silence-anchored batches: the chain kernels (k_dp5_any set 2, k_dp4x) on the head's auxiliary lane and the
# throughput kernels (k_dp5_any set 1, narrow pieces, k_mix) on the caller's stream (BFA_SPLIT_LAYOUT=1, experiment) against the layout as it is
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in 1 2; do for v in 0 1; do
  export BFA_SPLIT_LAYOUT=$v
  for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 16 2>/dev/null | last | msof "split_layout=$v c5proxy peak $p"; done
  python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "split_layout=$v c5proxy 3 in flight"
done; done
