#!/bin/bash
# C5 proxy at full batch + the small-call latencies with the one-launch wide kernel (k_dp5_any); then pytest -m gpu
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
TAG=${1:-r6h}
mkdir -p gpurun_out/$TAG
python tools/latency_realtext.py > gpurun_out/$TAG/latency_realtext.txt 2>&1; grep "^{" gpurun_out/$TAG/latency_realtext.txt | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('B', d['B'], 'frames', d['frames'], 'device ms', round(d['device_ms_back_to_back'], 3), 'sync', round(d['ms_per_call_with_sync'], 3), d['extract_timestamps_from_logits_ms'])"
BFA_BS=16 BFA_DEVICE_ONLY=1 bash tools/timeline.sh ${TAG}_b16 2 python $ROOT/tools/latency_realtext.py > gpurun_out/$TAG/b16.txt 2>&1
grep -v rocclr gpurun_out/$TAG/b16.txt | grep -v "at::native" | tail -24
bash tools/timeline.sh ${TAG}_c5 2 python $ROOT/tools/softness.py --shapes c5proxy --peaks 9 --steps 3 --warmup 2 --parity 4 > gpurun_out/$TAG/c5.txt 2>&1
grep -v rocclr gpurun_out/tl_${TAG}_c5/timeline.txt | grep -v "at::native" | head -30
python tools/softness.py --shapes c5proxy,realtext --peaks 9,5,3 --steps 10 --out gpurun_out/$TAG/softness.jsonl > gpurun_out/$TAG/softness.log 2>&1
python tools/softness.py --shapes c5proxy --peaks 9,3 --steps 10 --wide-any-max -1 > gpurun_out/$TAG/c5_classkernels.txt 2>&1
python tools/softness.py --shapes c5proxy --peaks 9,3 --steps 10 --wide-any-max 1000000 > gpurun_out/$TAG/c5_allmerged.txt 2>&1
python -c "
import json
for f, tag in (('gpurun_out/$TAG/softness.jsonl', 'default (pieces merged) '), ('gpurun_out/$TAG/c5_classkernels.txt', 'class kernels (rounds 2-5)'), ('gpurun_out/$TAG/c5_allmerged.txt', 'slots merged as well    ')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(tag, d['shape'], d['peak'], 'ms %.3f' % d['ms_per_call'], d['parity'])"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -8 gpurun_out/$TAG/pytest_gpu.log
