#!/bin/bash
# tools/r6_widex.sh -- the wide fallbacks (Rw 6 / 8) of silence-anchored calls through the exact window: tests, soak, A/B (BFA_NO_WIDE_XWIN=1 = before)
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for s in 81 82 83; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
for rep in 1 2; do for v in 1 0; do
  export BFA_NO_WIDE_XWIN=$v
  for p in 9 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 32 2>/dev/null | last | msof "no_wide_xwin=$v c5proxy peak $p"; done
  for p in 9 5; do BFA_PEAK=$p BFA_BS=1,4,16,64 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('no_wide_xwin=$v peak $p B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'])"; done
done; done
unset BFA_NO_WIDE_XWIN
bash tools/timeline.sh r6w_b16 2 env BFA_BS=16 BFA_DEVICE_ONLY=1 python $PWD/tools/latency_realtext.py 2>&1 | grep -v "^W2026" > gpurun_out/r6w_b16_timeline.txt
tail -34 gpurun_out/r6w_b16_timeline.txt | cut -c1-120
