#!/usr/bin/env python3
"""tools/plan_stamps.py -- phase clocks of k_plan_seg (the planner of the silence-anchored mode) on the real-text batch (B = 4096,
T = 1000, S = 40, ph66 head from raw logits, one head per call): needs a library built with -DBFA_PLAN_STAMPS
(tools/build_variant.sh plan_stamps bfa_segment -DBFA_PLAN_STAMPS; run with BFA_HIP_LIBRARY=...).  Lane 0 of every planner
leaves s_memrealtime stamps (100 MHz) at the phase boundaries in the utterance's global scratch; the tool finds them in the
decoder's workspace by two magic words."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import synth_realtext  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402
from bournemouth_forced_aligner_amd.forced_alignment import align_heads  # noqa: E402

B, T, S = 4096, 1000, 40
dev = torch.device("cuda", 0)
xp, xg, tp, tg = synth_realtext(B, T, S, 2003, dev)
ap = AlignmentUtils(blank_id=66, silence_id=0)
hint = ap.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=True, n_classes=67)
Tl = torch.full((B,), T, dtype=torch.int32, device=dev)
Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
for _ in range(3):
    (res, stats), = align_heads([ap], [xp], [tp], Tl, Sl, class_masks=[hint])
torch.cuda.synchronize()
ws = ap.viterbi_decoder._ws.buf
w = ws[:ws.numel() // 4 * 4].view(torch.int32)
M0, M1 = np.int32(np.uint32(0xc3c3abcd)), np.int32(np.uint32(0x5a5a1234))
E0, E1 = np.int32(np.uint32(0xa5a5dcba)), np.int32(np.uint32(0x3c3c4321))
hit = torch.nonzero((w[:-24] == int(M0)) & (w[1:-23] == int(M1)) & (w[22:-2] == int(E0)) & (w[23:-1] == int(E1))).flatten()
rows = torch.stack([w[hit + k] for k in range(24)], dim=1).cpu().numpy().astype(np.int64) & 0xffffffff
st = [(rows[:, 2 + 2 * k] | (rows[:, 3 + 2 * k] << 32)) for k in range(8)]
na, npieces = rows[:, 18], rows[:, 19]
Tn, wg = rows[:, 20], rows[:, 21]
t0 = min(s.min() for s in st)
us = lambda x: (x - t0) / 100.0
names = ["stage P(SIL) in LDS (+ uT / uS / tokens)", "target SIL groups", "top-level silences (cumsum + windows)", "match + segments + validation",
         "item range (atomic)", "piece loop (sub-silences, anchors, items)", "list appends + end"]
print(f"planners with stamps: {len(Tn)}; pieces per utterance {npieces.mean():.1f}, audio silences {na.mean():.1f}; "
      f"first start 0, last start {us(st[0]).max():.1f} us, last end {us(st[7]).max():.1f} us")
tot = (st[7] - st[0]) / 100.0
print(f"per utterance: {tot.mean():.1f} us mean, {np.percentile(tot, 50):.1f} median, {tot.max():.1f} max")
for k, n in enumerate(names):
    d = (st[k + 1] - st[k]) / 100.0
    print(f"  {n:46s} {d.mean():7.2f} us mean  {np.percentile(d, 50):7.2f} median  {d.max():7.2f} max   {100 * d.mean() / tot.mean():5.1f} %")
first = {}
for i in np.argsort(st[0]):
    first.setdefault(int(wg[i]), []).append(i)
second = [v[1] for v in first.values() if len(v) > 1]
if second:
    gap = np.array([(st[0][v[1]] - st[7][v[0]]) / 100.0 for v in first.values() if len(v) > 1])
    print(f"workgroups with two utterances: {len(second)}; gap between the end of the first and the start of the second {gap.mean():.2f} us mean")
