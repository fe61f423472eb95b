#!/bin/bash
# tools/r3_q.sh -- realtext step time against hardware-queue count, steps in flight and the staging margin of k_postconf
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
ms() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'])"; }
for q in 4 8; do for fl in 1 2 3 4; do
  GPU_MAX_HW_QUEUES=$q python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight $fl 2>/dev/null | ms "queues $q inflight $fl"
done; done
for v in k8 k2; do for fl in 1 3; do
  BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight $fl 2>/dev/null | ms "variant $v inflight $fl"
done; done
for p in normal high; do
  BFA_HEAD_PRIO=$p python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 1 2>/dev/null | ms "head prio $p inflight 1"
  BFA_HEAD_PRIO=$p python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 3 2>/dev/null | ms "head prio $p inflight 3"
done
