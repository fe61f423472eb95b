#!/bin/bash
# tools/ktrace_cmd.sh <tag> <command...> -- rocprofv3 kernel trace (+stats) of an arbitrary command
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
python $ROOT/tools/prof_summary.py $OUT 2>&1 | grep -E "calls=" | head -30
