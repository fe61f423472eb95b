#!/bin/bash
# same box, interleaved: this tree against the tree of round 5 (tools/ubench/bin/r05tree, git archive 4c76cbb) on the headline,
# realtext (one call at a time / three in flight), C2, C4 shard
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OLD=$ROOT/tools/ubench/bin/r05tree
ms() { python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d.get('roofline') or {}; print('$1', 'ms/step %.4f' % d['ms_per_step'], 'K1 %.4f' % r['kernel_ms'] if r.get('kernel_ms') else '')"; }
for rep in 1 2; do
  for t in new old; do
    if [ $t = new ]; then D=$ROOT; else D=$OLD; fi
    (cd $D && python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "$t headline")
    (cd $D && python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | ms "$t realtext-1")
    (cd $D && python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | ms "$t realtext-3")
    (cd $D && python bench.py --steps 20 --warmup 5 --no-cpu --no-confidences 2>/dev/null | ms "$t headline-noconf")
    (cd $D && python bench.py --steps 20 --warmup 5 --no-cpu --inflight 1 2>/dev/null | ms "$t headline-inflight1")
    (cd $D && python bench.py --config c2 --steps 50 --warmup 10 --no-cpu --inflight 1 2>/dev/null | ms "$t c2-inflight1")
    (cd $D && python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 16 2>/dev/null | ms "$t c4shard")
  done
done
