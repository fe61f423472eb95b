#!/usr/bin/env python3
"""tools/mix_times.py -- per-workgroup timeline of k_mix on one unsorted mixed-length call, from a library built with
-DBFA_MIX_TIMES (tools/build_variant.sh mixt bfa_dp_nk5_p9 -DBFA_MIX_TIMES; BFA_HIP_LIBRARY=.../libbfa_mixt.so): start
offset, DP time and walk time by utterance length."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
lp, tk, Tl, Sl = synth.synth_ragged(B, 200, 3000, 67, 7, dev)
T = Tl.cpu().numpy() if torch.is_tensor(Tl) else np.asarray(Tl)
S = Sl.cpu().numpy() if torch.is_tensor(Sl) else np.asarray(Sl)
au = AlignmentUtils(66, 0, silence_anchors=10)
hint = au.viterbi_decoder.class_mask_hint(T, S, has_sil=False, n_classes=67)
for _ in range(4):
    res = au.viterbi_decoder.align_batch(lp, tk, Tl, Sl, class_mask=hint)
torch.cuda.synchronize()
st, md = res.status.cpu().numpy().astype(np.int64), res.mode.cpu().numpy().astype(np.int64)
start = (md - md.min()) * 0.01                    # us
dp = ((st >> 16) & 0xffff) * 0.04
tot = (st & 0xffff) * 0.04
print("workgroups %d, first start 0, last start %.1f us, last end %.1f us" % (B, start.max(), (start + tot).max()))
edges = [0, 400, 800, 1200, 1600, 2000, 2400, 2800, 3001]
print("   T range      n  start(mean/max)  dp_us(mean/max)  walk_us(mean/max)  end(max)  dp ns/frame")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (T >= lo) & (T < hi)
    if not m.any():
        continue
    print("%5d-%5d %6d  %7.1f %7.1f  %7.1f %7.1f   %7.1f %7.1f   %7.1f   %6.1f" % (
        lo, hi, m.sum(), start[m].mean(), start[m].max(), dp[m].mean(), dp[m].max(), (tot - dp)[m].mean(), (tot - dp)[m].max(),
        (start + tot)[m].max(), (dp[m] * 1e3 / T[m]).mean()))
# how many workgroups are alive over time
ends = start + tot
for t in np.linspace(0, ends.max(), 14):
    print("t=%7.1f us  alive %5d" % (t, int(((start <= t) & (ends > t)).sum())))
