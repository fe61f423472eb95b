#!/bin/bash
# tools/r5_rt.sh -- the silence-anchored path after fusing the planner into the row pass: parity tests, realtext numbers, timeline
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5rt; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "segment or sil or level2 or realtext or golden or pipeline or soak" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 2>$OUT/rt1.err | last > $OUT/realtext_inflight1.json
python bench.py --config realtext --steps 20 --warmup 5 2>$OUT/rt3.err | last > $OUT/realtext.json
python tests/sil_time.py 2>/dev/null | last > $OUT/sil.json
python - <<PY
import json
for f in ("realtext_inflight1","realtext","sil"):
    try: d=json.load(open("$OUT/%s.json"%f))
    except Exception as e: print(f,"unreadable",e); continue
    print(f, "ms_per_step", d.get("ms_per_step"), "parity", d.get("parity"), "frac", (d.get("roofline") or {}).get("frac"))
PY
bash tools/timeline.sh r5rt 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 > $OUT/realtext_timeline.txt 2>&1
head -20 $OUT/realtext_timeline.txt
for s in 21 22; do timeout 900 python tests/soak.py 100 $s 2>&1 | tail -1; done
python tools/pipeline_time.py 4096 2>/dev/null | grep "^{" | cut -c 100-330
