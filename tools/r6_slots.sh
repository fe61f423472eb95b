#!/bin/bash
# tools/r6_slots.sh -- full silence-anchored batches: residual full-layout slots in the pieces' launch (default) against one class
# kernel per R (BFA_KEEP_SLOT_KERNELS=1), one box; tests first
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for k in 1 0; do
  export BFA_KEEP_SLOT_KERNELS=$k
  for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 32 2>/dev/null | last | msof "keep_slot_kernels=$k c5proxy peak $p"; done
  python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "keep_slot_kernels=$k c5proxy 3 in flight"
done; done
unset BFA_KEEP_SLOT_KERNELS
for s in 93 94; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
bash tools/timeline.sh r6s_c5 2 python $PWD/bench.py --config c5proxy --steps 3 --warmup 2 --min-timed-steps 3 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r6s_c5_timeline.txt
tail -36 gpurun_out/r6s_c5_timeline.txt | cut -c1-120
