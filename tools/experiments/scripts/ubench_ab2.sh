#!/bin/bash
# tools/ubench/ab2.sh [-a "<bench args>"] [-r rounds] lib1.so lib2.so ... -- interleaved timing, median/min of K1 per build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 200 --warmup 20"; R=5
while [ "${1:0:1}" = "-" ]; do case $1 in -a) ARGS=$2; shift 2;; -r) R=$2; shift 2;; esac; done
mkdir -p $ROOT/gpurun_out/ab2; rm -f $ROOT/gpurun_out/ab2/*.json
for i in $(seq 1 $R); do
  for lib in "$@"; do
    n=$(basename $lib .so)
    BFA_HIP_LIBRARY=$ROOT/$lib python $ROOT/bench.py --no-cpu $ARGS > $ROOT/gpurun_out/ab2/${n}__$i.json 2>/dev/null
  done
done
python - <<PY
import json,glob,collections,statistics as st
d=collections.defaultdict(list)
for f in sorted(glob.glob("$ROOT/gpurun_out/ab2/*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); n=f.split("/")[-1].split("__")[0]
        r=j.get("roofline") or {}
        d[n].append((j["ms_per_step"], r.get("kernel_ms"), (r.get("kernel_ms_stats") or {}).get("median"), (r.get("kernel_ms_stats") or {}).get("min")))
    except Exception as e: print(f,"ERR",e)
for n,v in d.items():
    print(f"{n:28s} step med={st.median(x[0] for x in v):.4f} min={min(x[0] for x in v):.4f} | k1 mean-med={st.median(x[1] for x in v):.4f} med-med={st.median(x[2] for x in v):.4f} min={min(x[3] for x in v):.4f}  runs={[round(x[0],4) for x in v]}")
PY
