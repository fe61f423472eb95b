ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/silp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $ROOT/tests/sil_time.py 2>&1 | tail -3
rocprofv3 --kernel-trace --output-format csv -d $OUT/r -o t -- python $ROOT/tests/sil_time.py > $OUT/r.log 2>&1
python - <<PY
import csv,glob
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/r/*kernel_trace.csv")[0])) if "bfa" in r["Kernel_Name"]]
idx=max(i for i,r in enumerate(rows) if "k_plan(" in r["Kernel_Name"])
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n=r["Kernel_Name"].replace("void bfa::(anonymous namespace)::","").replace("bfa::","")[:40]
    print(f'{n:40s} start={(int(r["Start_Timestamp"])-t0)/1e3:8.1f} end={(int(r["End_Timestamp"])-t0)/1e3:8.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","")}')
PY
