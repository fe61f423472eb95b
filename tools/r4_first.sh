#!/bin/bash
# tools/r4_first.sh -- state of HEAD at the start of a session: GPU tests, the mixed-length workloads, timelines of one
# rank's C4 shard and of a lone long utterance (where does the time between the first k_plan and the last kernel go?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
bash tools/r4_mixed.sh _base
bash tools/timeline.sh shard 1 python $ROOT/bench.py --config c4 --global-batch 4096 --steps 3 --warmup 2 --parity-sample 0 --no-cpu | tail -40
bash tools/timeline.sh lone3000 1 python $ROOT/tools/one_long.py 1 3000 120 | tail -14
bash tools/timeline.sh ragged 1 python $ROOT/bench.py --ragged --steps 3 --warmup 2 --no-cpu | tail -40
