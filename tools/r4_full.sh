#!/bin/bash
# tools/r4_full.sh [tag] -- the GPU test suite, then every workload's time on the current build (one box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
last() { grep "^{" | tail -1; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'parity', (d.get('parity_sample') or {}).get('mismatching_utterances', d.get('parity_mismatching_utterances')), 'frac', (d.get('roofline') or {}).get('frac'), 'k1', (d.get('roofline') or {}).get('kernel_ms'))"; }
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last > $OUT/bench$1.json; show headline < $OUT/bench$1.json
python bench.py --inflight 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | last > $OUT/bench_if1$1.json; show headline_inflight1 < $OUT/bench_if1$1.json
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 2>/dev/null | last > $OUT/c4_shard4096$1.json; show shard < $OUT/c4_shard4096$1.json
python bench.py --ragged --steps 30 2>/dev/null | last > $OUT/ragged$1.json; show ragged < $OUT/ragged$1.json
python bench.py --config c4 --steps 10 --warmup 2 2>/dev/null | last > $OUT/c4$1.json; show c4 < $OUT/c4$1.json
python bench.py --config c2 --no-cpu 2>/dev/null | last > $OUT/c2$1.json; show c2 < $OUT/c2$1.json
python bench.py --config realtext --no-cpu 2>/dev/null | last > $OUT/realtext$1.json; show realtext < $OUT/realtext$1.json
