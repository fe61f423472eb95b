#!/bin/bash
# round 6, first GPU call: the softness sweep (tools/softness.py) on the build of round 5 + this round's host changes, then pytest -m gpu
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
mkdir -p gpurun_out/r6a
python tools/softness.py --shapes headline,mixed --peaks 9,8,7.5,7,6,5,3 --out gpurun_out/r6a/softness_std.jsonl > gpurun_out/r6a/softness_std.log 2>&1
python tools/softness.py --shapes realtext,c5proxy --peaks 9,7,6,5,3 --steps 20 --out gpurun_out/r6a/softness_heads.jsonl > gpurun_out/r6a/softness_heads.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest_gpu.log 2>&1
tail -3 gpurun_out/r6a/pytest_gpu.log
cat gpurun_out/r6a/softness_std.jsonl gpurun_out/r6a/softness_heads.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print(d['shape'], d['peak'], 'ms %.3f' % d['ms_per_call'], 'frac %.3f' % d['hbm_frac'], 'lpf', d.get('path_logp_per_frame'), 'dead', d.get('sample_share_at_sentinel'), d['items'], d['status_ok'], d['parity'])
"
tail -5 gpurun_out/r6a/softness_std.log gpurun_out/r6a/softness_heads.log
