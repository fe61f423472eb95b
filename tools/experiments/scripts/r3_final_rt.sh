ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
python bench.py --config realtext --steps 20 --warmup 5 2>$OUT/realtext.err | last > $OUT/realtext.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --hw-queues 4 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1_q4.json
python tests/sil_time.py 2>/dev/null | last > $OUT/sil.json
python tools/pipeline_time.py 4096 2>/dev/null | grep "^{" > $OUT/pipeline.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_realtext -o t -- python $ROOT/bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 > $OUT/st_realtext.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_sil -o t -- python $ROOT/tests/sil_time.py > $OUT/st_sil.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_pipeline -o t -- python $ROOT/tools/pipeline_time.py 4096 1 > $OUT/st_pipeline.log 2>&1
cd $ROOT
for d in st_realtext st_sil st_pipeline; do cp $(find $OUT/$d -name "*kernel_stats.csv" | head -1) $OUT/${d}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$d; done
bash tools/timeline.sh r3f_realtext 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 > $OUT/realtext_timeline.txt 2>&1
bash tools/r3_pmc.sh rt2 python $ROOT/bench.py --config realtext --steps 3 --warmup 1 --settle-ms 0 --min-timed-steps 3 --parity-sample 0 --inflight 1 > $OUT/realtext_pmc.txt 2>&1
python - <<PY
import json
for f in ("realtext","realtext_inflight1","realtext_inflight1_q4","sil"):
    d=json.loads(open("$OUT/"+f+".json").read()); print(f, round(d["ms_per_step"],4))
for l in open("$OUT/pipeline.json"): d=json.loads(l); print(d["result"], round(d["ms_per_call_host_and_device"],2))
PY
