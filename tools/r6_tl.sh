#!/bin/bash
# round 6: kernel timelines of the soft / mixed real-text settings (tools/softness.py one setting per trace)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
mkdir -p gpurun_out/r6b
run() { # tag nplan shape peak
  bash tools/timeline.sh r6b_$1 $2 python $ROOT/tools/softness.py --shapes $3 --peaks $4 --steps 3 --warmup 2 --parity 4 > gpurun_out/r6b/$1.txt 2>&1
  tail -60 gpurun_out/r6b/$1.txt
}
run c5p9 2 c5proxy 9
run c5p5 2 c5proxy 5
run c5p3 2 c5proxy 3
run rt3 2 realtext 3
run hl6 1 headline 6
