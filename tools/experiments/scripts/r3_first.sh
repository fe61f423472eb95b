#!/bin/bash
# tools/r3_first.sh -- block 0 of the piece DPs through the unrolled path (the build) against the per-frame path (variant ff0:
# k_dp4_any of both head widths with -DBFA_FIRST_FAST=0), interleaved on one box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for i in 1 2 3; do for v in ${VARIANTS:-build ff0}; do
  if [ $v = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so; fi
  echo -n "$v: "; python tests/sil_time.py 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sil ms/step %.4f' % d['ms_per_step'], end='  ')"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 --inflight 1 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('realtext, one in flight %.4f' % d['ms_per_step'], end='  ')"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('three in flight %.4f' % d['ms_per_step'])"
done; done
