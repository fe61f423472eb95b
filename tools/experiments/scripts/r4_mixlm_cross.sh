#!/bin/bash
# tools/experiments/scripts/r4_mixlm_cross.sh -- where the lane-mask consumers of k_mix stop paying: mixed-length calls of 128 ... 4096
# utterances, the build (lane masks up to 256 utterances) against variants/libbfa_lmall.so (lane masks always), interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/cross.py <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bournemouth_forced_aligner_amd import AlignmentUtils
from tools import synth
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B in (128, 256, 384, 512, 1024, 2048, 4096):
    rng = np.random.default_rng(B)
    Tl = rng.integers(200, 3001, B).astype(np.int64); Sl = np.maximum(1, Tl // 25)
    lp, tk = synth.c4_utterances(np.arange(B), Tl, Sl, 67, 1004, dev)
    Td = torch.from_numpy(Tl.astype(np.int32)).to(dev); Sd = torch.from_numpy(Sl.astype(np.int32)).to(dev)
    hint = au.viterbi_decoder.class_mask_hint(Tl, Sl, has_sil=False, n_classes=67)
    fn = lambda: au.viterbi_decoder.align_batch(lp, tk, Td, Sd, class_mask=hint)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us/call", flush=True)
PY
for rep in 1 2; do
  echo "== build"; python /tmp/cross.py 2>&1 | grep "B="
  echo "== lmall"; BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_lmall.so python /tmp/cross.py 2>&1 | grep "B="
done
