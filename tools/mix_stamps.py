#!/usr/bin/env python3
"""tools/mix_stamps.py [B] -- per-workgroup clocks of k_mix on one unsorted mixed-length call (T ~ U[200, 3000], S = T // 25):
needs a library built with -DBFA_MIX_STAMPS (tools/build_variant.sh <name> bfa_dp_nk5_p9 -DBFA_MIX_STAMPS ...; run with
BFA_HIP_LIBRARY=...): every workgroup leaves its start / DP-end / walk-end s_memrealtime stamps (100 MHz) in the two spare
tuple rows of its utterance.  Prints the table of profiles/r04_mix_workgroup_timeline.txt."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import synth_ragged  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
lp, tk, Tl, Sl = synth_ragged(B, 200, 3000, 67, 9, dev)
au = AlignmentUtils(66, 0)
T, S = Tl.cpu().numpy().astype(np.int64), Sl.cpu().numpy().astype(np.int64)
hint = au.viterbi_decoder.class_mask_hint(T, S, has_sil=False, n_classes=67)
cap = int(S.max()) + 4
for _ in range(3):
    res = au.decode_alignments_device(lp, tk, Tl.to(dev), Sl.to(dev), class_mask=hint, seg_cap=cap)
torch.cuda.synchronize()
rows = res.segs[:, cap - 2:cap].cpu().numpy().reshape(B, 8).astype(np.int64)
u = lambda lo, hi: (lo & 0xffffffff) | ((hi & 0xffffffff) << 32)
st0, st1, st2 = u(rows[:, 0], rows[:, 1]), u(rows[:, 2], rows[:, 3]), u(rows[:, 4], rows[:, 5])
ok = rows[:, 6] == T
print(f"# workgroups with stamps: {int(ok.sum())} of {B} (items k_mix has no body for keep their class kernels)")
t0 = st0[ok].min()
us = lambda x: (x - t0) / 100.0   # s_memrealtime: 100 MHz
start, dpe, end = us(st0), us(st1), us(st2)
print(f"workgroups {int(ok.sum())}, first start 0, last start {start[ok].max():.1f} us, last end {end[ok].max():.1f} us")
print("   T range      n  start(mean/max)  dp_us(mean/max)  walk_us(mean/max)  end(max)  dp ns/frame  walk ns/frame")
for lo in range(0, 3000, 400):
    m = ok & (T >= lo) & (T < lo + 400 + (1 if lo + 400 >= 3000 else 0))
    if not m.any():
        continue
    dp, wk = dpe[m] - start[m], end[m] - dpe[m]
    print(f"{lo:5d}-{lo + 400:5d} {int(m.sum()):6d} {start[m].mean():8.1f} {start[m].max():7.1f} {dp.mean():8.1f} {dp.max():7.1f} "
          f"{wk.mean():9.1f} {wk.max():7.1f} {end[m].max():9.1f} {1e3 * (dp / T[m]).mean():9.1f} {1e3 * (wk / T[m]).mean():9.1f}")
edges = np.linspace(0, end[ok].max(), 14)
for e in edges:
    print(f"t={e:7.1f} us  alive {int(((start[ok] <= e) & (end[ok] > e)).sum()):5d}")
