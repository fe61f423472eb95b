#!/bin/bash
# tools/ubench/ab.sh A.so B.so [bench args] -- interleaved A/B timing of two builds of the library on one box
# (clocks drift with temperature, so only interleaved runs on the same box compare).
A=$1; B=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/ab
for i in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    BFA_HIP_LIBRARY=$lib python $ROOT/bench.py --no-cpu "$@" > $ROOT/gpurun_out/ab/${v}_$i.json 2>/dev/null
  done
done
python $ROOT/tools/ubench/extract.py $ROOT/gpurun_out/ab/A_*.json $ROOT/gpurun_out/ab/B_*.json
