#!/bin/bash
# tools/ubench/abn.sh [-a "<bench args>"] lib1.so lib2.so ... -- interleaved timing of several builds (3 rounds)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=""
if [ "$1" = "-a" ]; then ARGS=$2; shift 2; fi
mkdir -p $ROOT/gpurun_out/abn; rm -f $ROOT/gpurun_out/abn/*.json
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    BFA_HIP_LIBRARY=$lib python $ROOT/bench.py --no-cpu $ARGS > $ROOT/gpurun_out/abn/${n}_$i.json 2>/dev/null
  done
done
python $ROOT/tools/ubench/extract.py $ROOT/gpurun_out/abn/*.json
