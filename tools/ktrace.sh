#!/bin/bash
# tools/ktrace.sh <tag> [bench args] -- kernel trace only (fast): per-kernel average durations of bench.py
TAG=${1:-kt}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $OUT/trace.log 2>&1
python $ROOT/tools/prof_summary.py $OUT 2>&1 | grep -E "calls=" | grep -E "bfa" 
