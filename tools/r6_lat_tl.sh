#!/bin/bash
# the reference's call shape (B = 16 / 1, both heads, SIL in the targets): latencies, kernel timelines of ONE call; then the C5 proxy
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
TAG=${1:-r6f}
mkdir -p gpurun_out/$TAG
python tools/latency_realtext.py > gpurun_out/$TAG/latency_realtext.txt 2>&1; grep "^{" gpurun_out/$TAG/latency_realtext.txt | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('B', d['B'], 'frames', d['frames'], 'device ms', round(d['device_ms_back_to_back'], 3), 'sync', round(d['ms_per_call_with_sync'], 3), d['extract_timestamps_from_logits_ms'])"
for B in 16 1; do
  BFA_BS=$B BFA_DEVICE_ONLY=1 bash tools/timeline.sh ${TAG}_b$B 2 python $ROOT/tools/latency_realtext.py > gpurun_out/$TAG/b$B.txt 2>&1
  grep -v rocclr gpurun_out/$TAG/b$B.txt | grep -v "at::native" | tail -32
done
python tools/softness.py --shapes c5proxy,realtext --peaks 9,3 --steps 20 --out gpurun_out/$TAG/softness.jsonl > gpurun_out/$TAG/softness.log 2>&1
python -c "
import json
for ln in open('gpurun_out/$TAG/softness.jsonl'):
    d=json.loads(ln); print(d['shape'], d['peak'], 'ms %.3f' % d['ms_per_call'], d['parity'])"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -8 gpurun_out/$TAG/pytest_gpu.log
