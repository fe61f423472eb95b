#!/bin/bash
# tools/r6_check.sh -- the build as it is: GPU tests, soak, the real-text family of bench lines, the C5-proxy timeline
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for s in 97 98; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 32 2>/dev/null | last | msof "c5proxy peak $p"; done
python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 --parity-sample 0 2>/dev/null | last | msof "c5proxy 3 in flight"
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 32 2>/dev/null | last | msof "realtext inflight1"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | msof "realtext 3 in flight"
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | msof "headline"
python bench.py --ragged --steps 30 2>/dev/null | last | msof "ragged"
BFA_BS=1,16,64 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'])"
bash tools/timeline.sh r6u_c5 2 python $PWD/bench.py --config c5proxy --steps 3 --warmup 2 --min-timed-steps 3 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r6u_c5_timeline.txt
tail -38 gpurun_out/r6u_c5_timeline.txt | cut -c1-120
