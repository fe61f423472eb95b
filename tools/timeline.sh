#!/bin/bash
# tools/timeline.sh <tag> <k_plan launches per step> <command...> -- rocprofv3 kernel trace of a command; prints the
# kernel timeline (us from the first k_plan) of its LAST step and writes it + the kernel stats to gpurun_out/tl_<tag>/.
TAG=$1; NPLAN=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/cmd.log 2>&1
python - <<PY | tee $OUT/timeline.txt
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
plans = [i for i, r in enumerate(rows) if "k_plan(" in r["Kernel_Name"]]
i0 = plans[-$NPLAN]
t0 = int(rows[i0]["Start_Timestamp"])
tend = 0
for r in rows[i0:]:
    n = r["Kernel_Name"].replace("void bfa::(anonymous namespace)::", "").replace("void bfa::", "").replace("bfa::", "")[:46]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    tend = max(tend, e)
    print(f'{n:46s} start={s / 1e3:8.1f} end={e / 1e3:8.1f} dur={(e - s) / 1e3:8.1f} q={r.get("Queue_Id", "")}')
print(f"# last step: {tend / 1e3:.1f} us from the first k_plan to the last kernel end")
PY
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
tail -2 $OUT/cmd.log
