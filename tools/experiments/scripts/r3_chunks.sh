#!/bin/bash
# realtext step as N chunk calls (does the second read of a chunk's logits -- K1 after K0, then the probes -- hit the MALL?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
last() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['host_issue_ms_per_step'])"; }
for nf in 3 6; do for ch in 1 2 4 8 16; do
  echo -n "inflight $nf chunks $ch: "; python bench.py --config realtext --steps 20 --warmup 5 --inflight $nf --chunks $ch --parity-sample 0 --min-timed-steps 60 2>/dev/null | last
done; done
