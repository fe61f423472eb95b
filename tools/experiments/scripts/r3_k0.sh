#!/bin/bash
# tools/r3_k0.sh -- after a change to K0 (k_silprob3): the tests of the silence-anchored mode, then its timings
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
python tests/sil_time.py 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sil ms/step %.4f' % d['ms_per_step'], d['parity_sample'])"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 128 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('realtext ms/step %.4f' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 1 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('realtext, one in flight ms/step %.4f' % d['ms_per_step'])"
done
timeout 600 python tests/soak.py 100 31 2>&1 | tail -1
