#!/bin/bash
# tools/r3_pmc.sh <tag> <command...> -- rocprofv3 PMC passes (counters only, each in its own run) of a command; per-kernel
# means into gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o p -- "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o p -- "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o p -- "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o p -- "$@" > $OUT/pmc4.log 2>&1
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
