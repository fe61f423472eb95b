#!/bin/bash
# tools/r5_rt_bound.sh -- what bounds K1 of the silence-anchored path (k_dp4_any): the number of pieces or the longest chain?
# The real-text step one call at a time at (batch, frames, tokens) = (4096, 1000, 40), half the batch, half the length; the
# kernel timeline of the last step of each (tools/timeline.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "4096 1000 40" "2048 1000 40" "1024 1000 40" "4096 500 20" "8192 500 20"; do
  set -- $cfg
  echo "== batch $1 frames $2 tokens $3"
  bash tools/timeline.sh rtb_$1_$2 2 python $PWD/bench.py --config realtext --batch $1 --frames $2 --tokens $3 --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^{" | cut -c1-120
done
