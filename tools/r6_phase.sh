#!/bin/bash
# tools/r6_phase.sh -- fronts of both heads enqueued before any head's class kernels: tests, real text, C5 proxy (+ timeline), B = 16
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or {}).get('mismatching_utterances'))"; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 64 2>/dev/null | last | msof "realtext inflight1"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | msof "realtext 3 in flight"
for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 2>/dev/null | last | msof "c5proxy peak $p"; done
done
python bench.py --config c5proxy --steps 10 --warmup 3 --inflight 3 2>/dev/null | last | msof "c5proxy 3 in flight"
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | msof "headline"
BFA_BS=1,4,16,64 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'], d['extract_timestamps_from_logits_ms'])"
bash tools/timeline.sh r6p_c5 2 python $PWD/bench.py --config c5proxy --steps 3 --warmup 2 --min-timed-steps 3 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r6p_c5_timeline.txt
tail -45 gpurun_out/r6p_c5_timeline.txt | cut -c1-120
