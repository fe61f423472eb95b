#!/usr/bin/env python3
"""tools/latency_time.py -- per-call latency of small batches (the reference's process_sentence regime): the C-ABI
call alone (device-resident inputs, results left on the device) and AlignmentUtils.decode_alignments with its
list-of-tuples result."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402

dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((1, 75, 7), (1, 1000, 40), (16, 1000, 40), (1, 3000, 120)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl, Sl = [T] * B, [S] * B
    for name, fn in (("align_batch (device results)", lambda: au.viterbi_decoder.align_batch(lp, toks, Tl, Sl)),
                     ("decode_alignments (tuples)", lambda: au.decode_alignments(lp, toks, Tl, Sl))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:3d} T={T:5d} S={S:4d} {name:32s} {dt * 1e6:8.1f} us per call")
