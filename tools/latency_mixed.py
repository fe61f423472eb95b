#!/usr/bin/env python3
"""tools/latency_mixed.py -- small MIXED-length calls (the reference's process_sentences_batch regime: a handful of sentences of
different lengths per call), everything device-resident, class hint computed once: time per call back to back.
BFA_MIXMIN=n sets the Python side's copy of the library's MIX_MIN_BATCH (for a variant library built with another value)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
from tools import synth
if os.environ.get("BFA_MIXMIN"):
    _lib.MIX_MIN_BATCH = int(os.environ["BFA_MIXMIN"])
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for tmax in (1500, 3000):
    for B in (2, 3, 4, 8, 16, 32, 48, 64, 96):
        rng = np.random.default_rng(B * 7 + tmax)
        Tl = rng.integers(200, tmax + 1, B).astype(np.int64); Sl = np.maximum(1, Tl // 25)
        lp, tk = synth.c4_utterances(np.arange(B), Tl, Sl, 67, 1004, dev)
        Td = torch.from_numpy(Tl.astype(np.int32)).to(dev); Sd = torch.from_numpy(Sl.astype(np.int32)).to(dev)
        hint = au.viterbi_decoder.class_mask_hint(Tl, Sl, has_sil=False, n_classes=67)
        fn = lambda: au.viterbi_decoder.align_batch(lp, tk, Td, Sd, class_mask=hint)
        for _ in range(5): r = fn()
        torch.cuda.synchronize()
        assert int((r.status != 0).sum()) == 0
        t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize()
        print(f"T<={tmax} B={B}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us/call  ({int(Tl.sum())} frames, longest {int(Tl.max())})", flush=True)
