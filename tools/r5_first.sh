#!/bin/bash
# tools/r5_first.sh -- first gpurun call of round 5: GPU tests + smoke, headline line with the new keys, the gather legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5a; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | last > $OUT/bench.json
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 2>$OUT/c4s.err | last > $OUT/c4_shard4096.json
python bench.py --config c4 --steps 10 --warmup 2 2>$OUT/c4.err | last > $OUT/c4.json
python tools/api_time.py 2>$OUT/api.err | last > $OUT/api.json
python tools/pipeline_time.py 4096 2>$OUT/pipe.err | grep "^{" > $OUT/pipeline.json
python - <<PY
import json
for f in ("bench","c4_shard4096","c4","api"):
    try:
        d=json.load(open("$OUT/%s.json"%f))
    except Exception as e:
        print(f, "unreadable", e); continue
    keep={k:d.get(k) for k in ("value","ms_per_step","gather_ms","value_with_gather","gather","timed_steps_total","decode_alignments_ms","decode_alignments_lazy_ms","decode_alignments_device_ms","to_lists_ms") if k in d}
    r=d.get("roofline") or {}
    keep.update({k:r.get(k) for k in ("frac","kernel_ms","measured_copy_peak","frac_of_measured","valu") if k in r})
    if "parity_sample" in d: keep["parity"]=d["parity_sample"]
    print(f, json.dumps(keep))
PY
cat $OUT/pipeline.json | cut -c1-300
