#!/bin/bash
# tools/r3_far_ab.sh -- the far-block form of the split consumers (the build) against -DBFA_DP5_FAR=0 (variant far0), one box, interleaved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 %.4f' % d['ms_per_step'], end='  ')"; }
for i in 1 2 3; do for v in ${VARIANTS:-build far0}; do
  if [ $v = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so; fi
  echo -n "$v: "
  python bench.py --ragged --steps 30 2>/dev/null | j ragged
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 0 2>/dev/null | j "c4 shard"
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 0 2>/dev/null | j "c4"
  echo
done; done
