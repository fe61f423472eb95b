#!/bin/bash
# tools/r6_post.sh -- k_postconf after the staging fix: parity tests of the post-DP stages, phase stamps, the real-text step,
# the C5 proxy and the reference's call shape
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or {}).get('mismatching_utterances'))"; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
V=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_post_times.so
BFA_HIP_LIBRARY=$V python tools/post_stamps.py 2>&1 | tail -9
BFA_HIP_LIBRARY=$V python tools/post_stamps.py --ragged 2>&1 | tail -9
BFA_HIP_LIBRARY=$V python tools/post_stamps.py --ragged --peak 3 2>&1 | tail -9
for rep in 1 2; do
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 64 2>/dev/null | last | msof "realtext inflight1"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | msof "realtext 3 in flight"
done
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | msof "headline"
for p in 9 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 2>/dev/null | last | msof "c5proxy peak $p"; done
BFA_BS=16,64 python tools/latency_realtext.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B', d['B'], 'device ms %.3f' % d['device_ms_back_to_back'], d['extract_timestamps_from_logits_ms'])"
for s in 71 72; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
