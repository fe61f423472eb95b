#!/bin/bash
# tools/experiments/scripts/r4_c4_layouts.sh -- BASELINE config 4 at N = 1: how the 32768 utterances are handed to the library
# (calls of --chunk utterances, --halves sub-shards side by side on their own streams, --inflight steps in flight)
cd ${GRAFT_REPO_ROOT:-/root/repo}
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], (d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 2>/dev/null | j "chunk 16384"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --chunk 32768 2>/dev/null | j "chunk 32768"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --chunk 8192 2>/dev/null | j "chunk 8192"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --halves 2 2>/dev/null | j "halves 2"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --halves 2 --chunk 8192 2>/dev/null | j "halves 2 chunk 8192"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --halves 4 --chunk 8192 2>/dev/null | j "halves 4 chunk 8192"
python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 --inflight 2 2>/dev/null | j "inflight 2"
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 --halves 2 2>/dev/null | j "shard halves 2"
