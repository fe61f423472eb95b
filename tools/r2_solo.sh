ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/solo; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
prof() { # name, bench args
  n=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$n -o t -- python $ROOT/bench.py --ragged --steps 3 --no-cpu "$@" > $OUT/$n.log 2>&1
  echo "== $n: $@"
  python - <<PY
import csv,glob
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/$n/*kernel_trace.csv")[0])) if "bfa" in r["Kernel_Name"]]
idx=max(i for i,r in enumerate(rows) if "k_plan" in r["Kernel_Name"])
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print(f'{r["Kernel_Name"][:58]:58s} start={(int(r["Start_Timestamp"])-t0)/1e3:8.1f} end={(int(r["End_Timestamp"])-t0)/1e3:8.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q={r.get("Queue_Id","")}')
PY
}
prof t3000 --batch 64 --tlo 2990 --thi 3000
prof t2400 --batch 64 --tlo 2390 --thi 2400
prof t1600 --batch 64 --tlo 1590 --thi 1600
prof t1500 --batch 64 --tlo 1490 --thi 1500
prof c8 --batch 880 --tlo 2400 --thi 3000
