#!/bin/bash
# tools/r6_conf.sh -- cooperative confidences of long tuples: tests, stamps, C5 proxy / real text by sharpness, headline
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
msof() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.4f ms' % d['ms_per_step'], (d.get('parity') or d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
V=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_post_times.so
BFA_HIP_LIBRARY=$V python tools/post_stamps.py --ragged --peak 3 2>&1 | tail -9
BFA_HIP_LIBRARY=$V python tools/post_stamps.py --peak 3 2>&1 | tail -9
for rep in 1 2; do
for p in 9 6 3; do python bench.py --config c5proxy --peak $p --steps 10 --warmup 3 --parity-sample 32 2>/dev/null | last | msof "c5proxy peak $p"; done
for p in 9 3; do python bench.py --config realtext --peak $p --steps 20 --warmup 5 --inflight 1 --parity-sample 32 2>/dev/null | last | msof "realtext inflight1 peak $p"; done
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | msof "headline"
done
for s in 91 92; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
