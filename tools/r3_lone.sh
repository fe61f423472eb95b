#!/bin/bash
# tools/r3_lone.sh -- the lone-chain regime (k_one; one utterance, C2): is the chain waiting for its scalar mask stores?
# variants of the k_one translation unit: emission prefetch distance D = 2 / 4 / 6 frames, and no mask stores at all
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for v in ${VARIANTS:-d2 nostore noproduce noconsume}; do
  echo "== $v"; BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so python tools/latency_device.py 2>&1 | grep -v "^W20\|amdgpu.ids"
done
