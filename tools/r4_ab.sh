#!/bin/bash
# tools/r4_ab.sh [variant.so] -- one box: a lone long utterance, one rank's C4 shard, C4 at N = 1 (with a variant library: interleaved A/B)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
V=$ROOT/bournemouth-forced-aligner_amd/variants/$1
ms() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], 'parity', (d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
one() { python - <<PY
import os, sys, time, torch
sys.path.insert(0, "$ROOT")
import bench
from bournemouth_forced_aligner_amd import AlignmentUtils
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((1, 3000, 120), (1, 2000, 80), (64, 3000, 120)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl = torch.full((B,), T, dtype=torch.int32, device=dev); Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    hint = au.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=67)
    for _ in range(5): au.viterbi_decoder.align_batch(lp, toks, Tl, Sl, class_mask=hint)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): au.viterbi_decoder.align_batch(lp, toks, Tl, Sl, class_mask=hint)
    torch.cuda.synchronize(); print("$1 B=%d T=%d: %.3f ms per call" % (B, T, (time.perf_counter() - t0) / 20 * 1e3))
PY
}
LIBS="base"; [ -n "$1" ] && LIBS="base var"
for rep in 1 2; do
  for lib in $LIBS; do
    if [ $lib = var ]; then export BFA_HIP_LIBRARY=$V; else unset BFA_HIP_LIBRARY; fi
    one $lib
    python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 2>/dev/null | ms "$lib shard"
    python bench.py --ragged --steps 30 2>/dev/null | ms "$lib ragged"
    python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 2>/dev/null | ms "$lib c4"
  done
done
