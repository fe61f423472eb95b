#!/bin/bash
# tools/r6_narrow.sh -- full silence-anchored batches: the narrow pieces' kernel (k_dp4_any) behind the wide one on the caller's
# stream (default) against an auxiliary lane (BFA_NARROW_LANE=0/1/2, experiment build), one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in 1 2; do for nl in -1 0 1 2; do
  export BFA_NARROW_LANE=$nl
  for p in 9 6; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 0 2>/dev/null | last | msof "narrow_lane=$nl c5proxy peak $p"; done
  python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "narrow_lane=$nl c5proxy 3 in flight"
done; done
export BFA_NARROW_LANE=2
bash tools/timeline.sh r6n_c5 2 python $PWD/bench.py --config c5proxy --steps 3 --warmup 2 --min-timed-steps 3 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r6n_c5_timeline_lane2.txt
tail -42 gpurun_out/r6n_c5_timeline_lane2.txt | cut -c1-120
