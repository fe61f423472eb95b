ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c4p; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r -o t -- python $ROOT/bench.py --config c4 --steps 2 --warmup 1 --chunk ${1:-8192} --parity-sample 16 > $OUT/r.log 2>&1
python - <<PY
import csv,glob
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/r/*kernel_trace.csv")[0])) if "bfa" in r["Kernel_Name"]]
plans=[i for i,r in enumerate(rows) if "k_plan(" in r["Kernel_Name"]]
n=len(plans); per=n//3
st=plans[-per]
t0=int(rows[st]["Start_Timestamp"])
for r in rows[st:]:
    n=r["Kernel_Name"].replace("void bfa::(anonymous namespace)::","").replace("bfa::","")[:34]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if d>20: print(f'{n:34s} start={(int(r["Start_Timestamp"])-t0)/1e3:9.1f} end={(int(r["End_Timestamp"])-t0)/1e3:9.1f} dur={d:8.1f} q={r.get("Queue_Id","")}')
PY
