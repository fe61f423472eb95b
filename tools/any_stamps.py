#!/usr/bin/env python3
"""tools/any_stamps.py -- per-item clocks of k_dp4_any (K1 of the silence-anchored mode) on the real-text batch (B = 4096,
T = 1000, S = 40, ph66 head from raw logits, one head per call so that the kernel has the machine): needs a library built with
-DBFA_ANY_STAMPS (tools/build_variant.sh any_stamps bfa_dp_nk5_p6 -DBFA_ANY_STAMPS; run with BFA_HIP_LIBRARY=...): the
consumer wave of every item leaves its DP-start / DP-end / walk-end s_memrealtime stamps (100 MHz) in its own backpointer block (dead
by then; the tool finds them in the decoder's workspace by their two magic words).  Prints profiles/r05_any_item_timeline.txt."""
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
hit = torch.nonzero(((w[:-10] & -65536) == 0x5a5a0000) & (w[10:] == 0x3c3c3c3c)).flatten()
rows = torch.stack([w[hit + k] for k in range(10)], dim=1).cpu().numpy().astype(np.int64)
u = lambda lo, hi: (lo & 0xffffffff) | ((hi & 0xffffffff) << 32)
st0, st1, st2 = u(rows[:, 1], rows[:, 2]), u(rows[:, 3], rows[:, 4]), u(rows[:, 5], rows[:, 6])
Ts, L, wg = rows[:, 7], rows[:, 8], rows[:, 9]
ok = (Ts > 0) & (Ts <= T) & (st1 >= st0) & (st2 >= st1) & (L >= 3) & (L <= 4 * S + 1)
st0, st1, st2, Ts, L, wg = st0[ok], st1[ok], st2[ok], Ts[ok], L[ok], wg[ok]
t0 = st0.min()
us = lambda x: (x - t0) / 100.0
start, dpe, end = us(st0), us(st1), us(st2)
print(f"items with stamps: {len(Ts)} ({int((wg < B).sum())} utterance slots, {int((wg >= B).sum())} pieces); frames {int(Ts.sum())}; "
      f"first DP start 0, last DP start {start.max():.1f} us, last walk end {end.max():.1f} us")
print("    Ts range      n   frames  start(mean/max)  dp_us(mean/max)  walk_us(mean/max)  end(max)  dp ns/frame  walk ns/frame")
for lo, hi in ((10, 50), (50, 100), (100, 200), (200, 300), (300, 500), (500, 800), (800, 1001)):
    m = (Ts >= lo) & (Ts < hi)
    if not m.any():
        continue
    dp, wk = dpe[m] - start[m], end[m] - dpe[m]
    print(f"{lo:5d}-{hi:5d} {int(m.sum()):6d} {int(Ts[m].sum()):8d} {start[m].mean():8.1f} {start[m].max():7.1f} {dp.mean():8.1f} {dp.max():7.1f} "
          f"{wk.mean():9.1f} {wk.max():7.1f} {end[m].max():9.1f} {1e3 * (dp / Ts[m]).mean():9.1f} {1e3 * (wk / Ts[m]).mean():9.1f}")
dp = dpe - start
A = np.stack([np.ones_like(Ts, dtype=np.float64), Ts.astype(np.float64)], axis=1)
coef, *_ = np.linalg.lstsq(A, dp, rcond=None)
print(f"DP time ~ {coef[0]:.2f} us + {1e3 * coef[1]:.1f} ns/frame (least squares over all items); walk: "
      f"{np.linalg.lstsq(A, end - dpe, rcond=None)[0][0]:.2f} us + {1e3 * np.linalg.lstsq(A, end - dpe, rcond=None)[0][1]:.1f} ns/frame")
print(f"sum of item busy time (DP + walk): {float((end - start).sum()) / 1e3:.1f} ms of consumer-wave time; DP only {float(dp.sum()) / 1e3:.1f} ms")
edges = np.linspace(0, end.max(), 16)
for e in edges:
    al = (start <= e) & (end > e)
    print(f"t={e:7.1f} us  items in DP or walk {int(al.sum()):5d}   (in DP {int(((start <= e) & (dpe > e)).sum()):5d})")
