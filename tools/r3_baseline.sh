#!/bin/bash
# tools/r3_baseline.sh -- one gpurun call: GPU tests + smoke, then every workload's JSON line into gpurun_out/r3/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
TAG=${1:-base}
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_$TAG.log
tail -5 $OUT/pytest_$TAG.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -c 600 $OUT/bench_$TAG.err
timeout 600 python bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu > $OUT/bench_inflight1_$TAG.json 2>> $OUT/bench_$TAG.err
timeout 600 python bench.py --config realtext --steps 20 --warmup 5 > $OUT/realtext_$TAG.json 2> $OUT/realtext_$TAG.err; tail -c 600 $OUT/realtext_$TAG.err
timeout 600 python bench.py --config c2 --steps 50 --warmup 10 --no-cpu --inflight 1 > $OUT/c2_$TAG.json 2> $OUT/c2_$TAG.err
timeout 900 python bench.py --config c4 --steps 10 --warmup 2 > $OUT/c4_$TAG.json 2> $OUT/c4_$TAG.err; tail -c 600 $OUT/c4_$TAG.err
timeout 600 python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 > $OUT/c4shard_$TAG.json 2> $OUT/c4shard_$TAG.err
timeout 600 python bench.py --ragged --steps 20 > $OUT/ragged_$TAG.json 2> $OUT/ragged_$TAG.err
timeout 600 python tools/pipeline_time.py 4096 > $OUT/pipeline_$TAG.log 2>&1
timeout 300 python tools/latency_time.py > $OUT/latency_$TAG.log 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*_$TAG.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "ms/step", round(d.get("ms_per_step",0),4), "value", d.get("value", d.get("frames_per_s")), "frac", (d.get("roofline") or {}).get("frac", d.get("hbm_frac")), "whole", (d.get("roofline") or {}).get("whole_step_frac"), "parity", d.get("parity") or d.get("parity_sample") or (d.get("cpu_baseline") or {}).get("parity_mismatching_utterances"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -3 $OUT/pipeline_$TAG.log $OUT/latency_$TAG.log
