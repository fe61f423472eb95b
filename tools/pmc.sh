#!/bin/bash
# tools/pmc.sh <tag> <command...> -- rocprofv3 PMC passes of a command (counters only, each set in its own run, no trace
# domains beside the kernel trace); per-kernel means into gpurun_out/pmc_<tag>/summary.txt.  Use absolute paths in <command>.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/pmc$i -o p -- "$@" > $OUT/pmc$i.log 2>&1
done
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
