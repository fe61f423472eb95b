#!/bin/bash
# tools/r5_check.sh -- one gpurun call on a fresh build: GPU tests, smoke, the default bench line, the real-text line.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5c; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | last | tee $OUT/bench.json | cut -c1-400
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last | tee $OUT/realtext_inflight1.json | cut -c1-300
python tools/api_time.py 2>/dev/null | last | tee $OUT/api.json | cut -c1-600
python tools/pipeline_time.py 4096 2>/dev/null | grep "^{" | tee $OUT/pipeline.json | cut -c1-800
