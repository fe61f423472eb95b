ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 900 python tests/soak.py 150 4242 2>&1 | tail -2
OUT=$ROOT/gpurun_out/k2f; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h -o t -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu --settle-ms 0 > $OUT/h.log 2>&1
python $ROOT/tools/prof_summary.py $OUT/h 2>/dev/null | grep -E "bfa" | head; python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$OUT/h/*kernel_stats.csv")[0])):
    if "bfa" in r["Name"]: print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
cd $ROOT
python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
python bench.py --ragged --no-cpu
python bench.py --config c4 --steps 5 --chunk 8192 2>/dev/null | cut -c1-330
