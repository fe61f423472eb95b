ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pipe; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python -m pytest $ROOT/tests -m gpu -q -x 2>&1 | tail -3
for f in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f$f -o t -- python $ROOT/tools/pipeline_time.py 4096 $f > $OUT/f$f.log 2>&1
  tail -1 $OUT/f$f.log
  python - <<PY
import csv,glob
f=glob.glob("$OUT/f$f/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    if "bfa" in r["Name"]:
        per=float(r["TotalDurationNs"])/3/1e3   # three calls
        tot+=per
        print(f'   {r["Name"][:70]:70s} calls={r["Calls"]:>4s} per-call us={per:9.1f}')
print("   fused=$f: bfa kernel time per pipeline call = %.3f ms"%(tot/1e3))
PY
done
