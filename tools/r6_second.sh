#!/bin/bash
# round 6: the soft / mixed settings after the fast window's early exit (doomed), the exact rerun's closed-form tails, the window
# routing by the handle's history and k_postconf's wide windows; then the bench lines, the call-shape latencies and pytest -m gpu
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
TAG=${1:-r6c}
mkdir -p gpurun_out/$TAG
python tools/softness.py --shapes headline --peaks 9,8,7.5,7,6,5,3 --routing 0 --out gpurun_out/$TAG/softness_unrouted.jsonl > gpurun_out/$TAG/softness.log 2>&1
python tools/softness.py --shapes headline --peaks 9,8,7.5,7,6,5,3 --out gpurun_out/$TAG/softness.jsonl >> gpurun_out/$TAG/softness.log 2>&1
python tools/softness.py --shapes mixed --peaks 9,7,5,3 --out gpurun_out/$TAG/softness.jsonl >> gpurun_out/$TAG/softness.log 2>&1
python tools/softness.py --shapes realtext,c5proxy --peaks 9,7,5,3 --steps 20 --out gpurun_out/$TAG/softness.jsonl >> gpurun_out/$TAG/softness.log 2>&1
cat gpurun_out/$TAG/softness_unrouted.jsonl gpurun_out/$TAG/softness.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print(d['shape'], d['peak'], 'routing', d.get('window_routing'), 'ms %.3f' % d['ms_per_call'], 'frac %.3f' % d['hbm_frac'], 'lpf', d.get('path_logp_per_frame'), 'dead', d.get('sample_share_at_sentinel'), d['items'], d['status_ok'], d['parity'])
"
tail -5 gpurun_out/$TAG/softness.log
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('headline ms/step', d['ms_per_step'], 'K1', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 > gpurun_out/$TAG/realtext1.json 2>> gpurun_out/$TAG/bench.err; python -c "
import json; d=json.load(open('gpurun_out/$TAG/realtext1.json')); print('realtext inflight 1 ms/step', d['ms_per_step'], d['parity'])"
python tools/latency_realtext.py > gpurun_out/$TAG/latency_realtext.txt 2>&1; cat gpurun_out/$TAG/latency_realtext.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -15 gpurun_out/$TAG/pytest_gpu.log
