#!/bin/bash
# kernel timelines of ONE bfa_align_heads call in the reference's call shape (B = 16 / 1, both heads, SIL in the targets)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
mkdir -p gpurun_out/r6e
for B in 16 1; do
  BFA_BS=$B BFA_DEVICE_ONLY=1 bash tools/timeline.sh r6e_b$B 2 python $ROOT/tools/latency_realtext.py > gpurun_out/r6e/b$B.txt 2>&1
  grep -v rocclr gpurun_out/r6e/b$B.txt | tail -50
done
BFA_PEAK=5 BFA_BS=16 BFA_DEVICE_ONLY=1 bash tools/timeline.sh r6e_b16p5 2 python $ROOT/tools/latency_realtext.py > gpurun_out/r6e/b16p5.txt 2>&1
grep -v rocclr gpurun_out/r6e/b16p5.txt | tail -40
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "soft or reference_post" 2>&1 | tail -5
python tools/softness.py --shapes headline --peaks 9,7.5,7 > gpurun_out/r6e/soft.txt 2>&1; python -c "
import json
for ln in open('gpurun_out/r6e/soft.txt'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['shape'], d['peak'], 'ms %.3f' % d['ms_per_call'], d['items'], d['parity'])"
