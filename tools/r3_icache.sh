#!/bin/bash
# tools/r3_icache.sh -- instruction-cache counters of the one-kernel small-batch path (one utterance per call, C2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "Counter_Name" | grep "SQ_" > $OUT/avail.txt
i=0
for c in ${COUNTER_SETS:-"SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"}; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc ${c//,/ } -d $OUT/p$i -o p -- python $ROOT/tools/latency_device.py > $OUT/p$i.log 2>&1
done
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bfa" not in n: continue
        n = n.replace("void bfa::(anonymous namespace)::", "").replace("void bfa::", "").replace("bfa::", "")[:44] + " grid=" + r.get("Grid_Size", "?")
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    cs = agg[k]
    print(k, "(n=%d)" % len(next(iter(cs.values()))))
    for c in sorted(cs):
        v = cs[c]
        print("    %-28s %16.0f" % (c, sum(v) / len(v)))
PY
grep -c . $OUT/avail.txt
