#!/bin/bash
# tools/r3_rt.sh [tag] -- realtext: parity-checked run, timeline of one step, step times for 1 / 2 / 3 steps in flight
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rt}
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT; cd $ROOT
ms() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'parity', (d.get('parity') or {}).get('mismatching_utterances'), (d.get('parity') or {}).get('confidence_beyond_1e-4'), 'segmented', d.get('segmented_utterances'))"; }
python bench.py --config realtext --steps 20 --warmup 5 2> $OUT/rt_$TAG.err | tee $OUT/realtext_$TAG.json | ms "realtext inflight1"
tail -3 $OUT/rt_$TAG.err
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 2 2>/dev/null | ms "realtext inflight2"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 3 2>/dev/null | ms "realtext inflight3"
python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --separate-post 2>/dev/null | ms "realtext separate-post"
for v in ${BFA_VARIANTS:-}; do
  BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 64 2>/dev/null | ms "variant $v inflight1"
  BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 2 2>/dev/null | ms "variant $v inflight2"
done
bash tools/timeline.sh rt_$TAG 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 > $OUT/timeline_$TAG.txt 2>&1
cat $OUT/timeline_$TAG.txt | grep -v "^W2026"
