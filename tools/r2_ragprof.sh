ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ragp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r -o t -- python $ROOT/bench.py --ragged --steps 3 --no-cpu > $OUT/r.log 2>&1
python - <<PY
import csv,glob
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/r/*kernel_trace.csv")[0])) if "bfa" in r["Kernel_Name"]]
# last step: take the last k_plan onwards
idx=max(i for i,r in enumerate(rows) if "k_plan" in r["Kernel_Name"])
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print(f'{r["Kernel_Name"][:58]:58s} start={(int(r["Start_Timestamp"])-t0)/1e3:8.1f} end={(int(r["End_Timestamp"])-t0)/1e3:8.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","")}')
PY
