#!/bin/bash
# tools/r6_mixclass.sh -- per window class: vector instructions per frame and kernel time of the fast window's class kernel, the
# exact window's class kernel and k_mix on 4096 x 1000-frame utterances of that class (tools/mix_class.py) -> profiles/r06_mix_class_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mixclass
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 20 40 80 115; do for mode in fast exact mix; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/${S}_$mode -o p -- python $ROOT/tools/mix_class.py --tokens $S --mode $mode > $OUT/${S}_$mode.log 2>&1
done; done
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections, json, os
print("# tools/r6_mixclass.sh: B = 4096, T = 1000, standard mode, C = 67, peak 9; per launch (mean over the launches of the run)")
print("# %-4s %-6s %-44s %10s %12s %10s" % ("S", "mode", "kernel", "us", "VALU insts", "per frame"))
for S in (20, 40, 80, 115):
    for mode in ("fast", "exact", "mix"):
        d = "$OUT/%d_%s" % (S, mode)
        agg = collections.defaultdict(list); dur = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == "SQ_INSTS_VALU" and "bfa" in r["Kernel_Name"]: agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "bfa" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        line = [l for l in open("$OUT/%d_%s.log" % (S, mode)) if l.startswith("{")]
        for k in sorted(agg, key=lambda k: -sum(agg[k]) / len(agg[k])):
            v = sum(agg[k]) / len(agg[k])
            if v < 1e6: continue
            n = k.replace("void bfa::(anonymous namespace)::", "").replace("void bfa::", "").replace("bfa::", "")[:44]
            u = dur.get(k, [0]); 
            print("  %-4d %-6s %-44s %10.1f %12.0f %10.1f" % (S, mode, n, sum(u) / len(u), v, v / 4096e3))
        if line: print("       ", line[-1].strip()[:200])
PY
cp $OUT/summary.txt $ROOT/gpurun_out/r06_mix_class_pmc.txt
