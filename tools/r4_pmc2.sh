#!/bin/bash
# tools/r4_pmc2.sh <tag> <command...> -- r3_pmc.sh's passes plus an instruction-cache pass, per-kernel means
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/r3_pmc.sh $TAG "$@" > /dev/null 2>&1
OUT=$ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d $OUT/pmc5 -o p -- "$@" > $OUT/pmc5.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH -d $OUT/pmc6 -o p -- "$@" > $OUT/pmc6.log 2>&1
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bfa" not in n: continue
        n = n.replace("void bfa::(anonymous namespace)::", "").replace("void bfa::", "").replace("bfa::", "")[:44]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    cs = agg[k]
    print(k, "(n=%d)" % len(next(iter(cs.values()))))
    for c in sorted(cs):
        v = cs[c]
        print("    %-26s %16.0f" % (c, sum(v) / len(v)))
PY
