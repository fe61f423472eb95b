#!/bin/bash
# tools/r3_rt_serial.sh <variants...> -- realtext with the heads one after the other (BFA_HEADS_SERIAL=1): every kernel's
# duration without another head beside it, per library variant ("main" = the built library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT; cd $ROOT
export BFA_HEADS_SERIAL=1
for v in "$@"; do
  if [ "$v" != "main" ]; then export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; else unset BFA_HIP_LIBRARY; fi
  bash tools/timeline.sh rts_$v 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 > $OUT/timeline_serial_$v.txt 2>&1
  echo "== $v"; grep -v "^W2026\|elementwise\|fillBuffer\|copyBuffer" $OUT/timeline_serial_$v.txt
done
