#!/usr/bin/env python3
"""tools/post_stamps.py -- where a k_postconf wave spends its cycles (A/B build with -DBFA_POST_TIMES:
`tools/build_variant.sh post_times bfa_post -DBFA_POST_TIMES`, run with BFA_HIP_LIBRARY=<that file>).  The kernel sums
clock64() deltas per phase over the utterances of its launches; this prints cycles per utterance and phase for the real-text
batch (4096 x 1000 x 40, both heads) or the C5 proxy (--ragged)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import synth_realtext, synth_realtext_ragged  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils, _lib  # noqa: E402
from bournemouth_forced_aligner_amd.forced_alignment import align_heads  # noqa: E402

PHASES = ["tuples in, coverage, order check", "windows (prefix sums)", "stage cells (loads + exp)", "segment means", "wide test",
          "passes", "confidences: alias test", "confidences"]
ap_ = argparse.ArgumentParser()
ap_.add_argument("--batch", type=int, default=4096)
ap_.add_argument("--peak", type=float, default=9.0)
ap_.add_argument("--ragged", action="store_true")
ap_.add_argument("--calls", type=int, default=10)
args = ap_.parse_args()
dev = torch.device("cuda", 0)
B = args.batch
if args.ragged:
    xs = synth_realtext_ragged(B, 300, 1870, 12, 2004, dev, peak=args.peak, gpeak=max(1.0, args.peak - 2.0))
    (xp, xg, tp, tg), Tl, Sl = xs[:4], xs[4].numpy().astype(np.int64), xs[5].numpy().astype(np.int64)
else:
    xp, xg, tp, tg = synth_realtext(B, 1000, 40, 2003, dev, peak=args.peak, gpeak=max(1.0, args.peak - 2.0))
    Tl, Sl = np.full(B, 1000, np.int64), np.full(B, 40, np.int64)
Td, Sd = torch.from_numpy(Tl.astype(np.int32)).to(dev), torch.from_numpy(Sl.astype(np.int32)).to(dev)
ap, ag = AlignmentUtils(blank_id=66, silence_id=0), AlignmentUtils(blank_id=16, silence_id=0)
vd = ap.viterbi_decoder
hints = [vd.class_mask_hint(Tl, Sl, has_sil=True, n_classes=67), vd.class_mask_hint(Tl, Sl, has_sil=True, n_classes=17)]
fn = lambda: align_heads([ap, ag], [xp, xg], [tp, tg], Td, Sd, class_masks=hints, post={"extend": True, "boundary_softness": 3})  # noqa: E731
lib = _lib.lib()
dump = lib.bfa_dbg_post_times
dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    fn()
torch.cuda.synchronize()
dump(buf, 1)
for _ in range(args.calls):
    fn()
torch.cuda.synchronize()
dump(buf, 0)
n = args.calls * B * 2
tot = sum(buf[k] for k in range(8))
print("k_postconf, cycles per utterance (both heads, %d calls of %d utterances, peak %g%s): total %.0f" %
      (args.calls, B, args.peak, ", ragged" if args.ragged else "", tot / n))
for k, name in enumerate(PHASES):
    print("  %-36s %9.0f  %5.1f %%" % (name, buf[k] / n, 100.0 * buf[k] / max(1, tot)))
