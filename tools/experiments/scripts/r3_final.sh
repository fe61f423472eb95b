#!/bin/bash
# tools/r3_final.sh -- one gpurun call: GPU tests + smoke, then every number quoted in README / DESIGN section 6 with the file
# behind it under gpurun_out/r3f/ (copied to profiles/r03_* afterwards), rocprofv3 kernel stats of the same commands.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | last > $OUT/bench.json
python bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu 2>/dev/null | last > $OUT/bench_inflight1.json
rm -f $OUT/bench_runs.jsonl; for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last >> $OUT/bench_runs.jsonl; done
python bench.py --config realtext --steps 20 --warmup 5 2>$OUT/realtext.err | last > $OUT/realtext.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --hw-queues 4 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1_q4.json
python bench.py --config c2 --steps 50 --warmup 10 --no-cpu --inflight 1 2>/dev/null | last > $OUT/c2.json
python bench.py --config c4 --steps 10 --warmup 2 2>/dev/null | last > $OUT/c4.json
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 2>/dev/null | last > $OUT/c4_shard4096.json
python bench.py --ragged --steps 30 2>/dev/null | last > $OUT/ragged.json
python tests/sil_time.py 2>/dev/null | last > $OUT/sil.json
python tools/pipeline_time.py 4096 2>/dev/null | grep "^{" > $OUT/pipeline.json
python tools/api_time.py 2>/dev/null | last > $OUT/api.json
python tools/latency_device.py 2>/dev/null > $OUT/latency.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_inflight1 -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu > $OUT/st_inflight1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_default -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu > $OUT/st_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_realtext -o t -- python $ROOT/bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 > $OUT/st_realtext.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_c4 -o t -- python $ROOT/bench.py --config c4 --steps 6 --warmup 2 --parity-sample 0 > $OUT/st_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_sil -o t -- python $ROOT/tests/sil_time.py > $OUT/st_sil.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_pipeline -o t -- python $ROOT/tools/pipeline_time.py 4096 1 > $OUT/st_pipeline.log 2>&1
cd $ROOT
for d in st_inflight1 st_default st_realtext st_c4 st_sil st_pipeline; do cp $(find $OUT/$d -name "*kernel_stats.csv" | head -1) $OUT/${d}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$d; done
bash tools/timeline.sh r3f_realtext 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 > $OUT/realtext_timeline.txt 2>&1
bash tools/timeline.sh r3f_ragged 1 python $ROOT/bench.py --ragged --steps 3 > $OUT/ragged_timeline.txt 2>&1
bash tools/profile.sh r3f > $OUT/profile.log 2>&1
bash tools/r3_pmc.sh rt_final python $ROOT/bench.py --config realtext --steps 3 --warmup 1 --settle-ms 0 --min-timed-steps 3 --parity-sample 0 --inflight 1 > $OUT/realtext_pmc_final.txt 2>&1
python tools/prof_summary.py gpurun_out/prof_r3f > $OUT/headline_summary.txt 2>&1
for s in 11 12 13; do timeout 900 python tests/soak.py 150 $s --record $OUT/soak.json 2>&1 | tail -1; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        for l in open(f):
            d=json.loads(l)
            print(os.path.basename(f), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ("ms_per_step","value","frames_per_s","hbm_frac","decode_alignments_ms","to_lists_ms","ms_per_call_host_and_device","total_utterances","total_mismatches")}, "frac", (d.get("roofline") or {}).get("frac"), "whole", (d.get("roofline") or {}).get("whole_step_frac"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
cat $OUT/latency.txt
