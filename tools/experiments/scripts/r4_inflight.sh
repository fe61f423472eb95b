#!/bin/bash
# tools/experiments/scripts/r4_inflight.sh -- headline step (alignment + confidence pass) against batches in flight
cd ${GRAFT_REPO_ROOT:-/root/repo}
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], 'alignment only %.4f' % d['alignment_only']['ms_per_step'], 'K1 %.4f' % d['roofline']['kernel_ms'])"; }
for rep in 1 2; do for n in 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu --inflight $n 2>/dev/null | j "inflight $n"; done; done
