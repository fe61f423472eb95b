#!/bin/bash
# tools/r4_mix.sh [tag] -- the one-kernel mixed-length path: its tests, then the mixed-length workloads and a timeline
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py -x -q -m gpu 2>&1 | tail -15
bash tools/r4_mixed.sh _mix$1
bash tools/timeline.sh ragged_mix$1 1 python $ROOT/bench.py --ragged --steps 3 --warmup 2 --no-cpu | grep -v "at::native\|rocclr\|rocprim" | tail -30
