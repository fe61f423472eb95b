#!/bin/bash
# tools/r3_c4.sh -- BASELINE config 4: one rank's shard (global batch 4096) and the whole batch at N = 1, against the
# number of sub-shards side by side, steps in flight and hardware queues
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
ms() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'parity', d['parity_sample']['mismatching_utterances'], 'pred', d.get('predicted_rank_ms'))"; }
for q in 4 8; do for hv in 1 2; do for fl in 1 2; do
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 --halves $hv --inflight $fl --hw-queues $q 2>/dev/null | ms "shard4096 q$q halves$hv inflight$fl"
done; done; done
for hv in 1 2; do for q in 4 8; do
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 --halves $hv --hw-queues $q 2>/dev/null | ms "c4 full q$q halves$hv"
done; done
