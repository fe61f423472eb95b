#!/bin/bash
# tools/r6_auxprio.sh -- the auxiliary streams (class kernels side by side) at another stream priority: queues of their own?
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in 1 2; do for pr in none hi lo; do
  if [ $pr = none ]; then unset BFA_AUX_PRIO; else export BFA_AUX_PRIO=$pr; fi
  python bench.py --config c5proxy --steps 10 --warmup 3 --parity-sample 0 2>/dev/null | last | msof "$pr c5proxy peak 9"
  python bench.py --config c5proxy --peak 3 --steps 10 --warmup 3 --parity-sample 0 2>/dev/null | last | msof "$pr c5proxy peak 3"
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last | msof "$pr realtext inflight1"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | msof "$pr realtext 3 in flight"
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | msof "$pr headline"
  python bench.py --ragged --steps 30 2>/dev/null | last | msof "$pr ragged"
  python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 0 2>/dev/null | last | msof "$pr c4"
  BFA_BS=16 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$pr B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'])"
done; done
export BFA_AUX_PRIO=hi
bash tools/timeline.sh r6q_c5 2 python $PWD/bench.py --config c5proxy --steps 3 --warmup 2 --min-timed-steps 3 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r6q_c5_timeline_hi.txt
tail -42 gpurun_out/r6q_c5_timeline_hi.txt | cut -c1-120
