#!/usr/bin/env python3
"""tools/latency_device.py -- small batches with everything device-resident (lengths, targets, class hint computed once):
host time to issue a call, time per call back to back, and time per call with a synchronisation after each
(B = 1 x T = 1000 -- the reference's process_sentence regime -- and BASELINE config 2)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bournemouth_forced_aligner_amd import AlignmentUtils
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((1, 1000, 40), (256, 600, 20)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl = torch.full((B,), T, dtype=torch.int32, device=dev); Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    hint = au.viterbi_decoder.class_mask_hint([T]*B, [S]*B, has_sil=False, n_classes=67)
    fn = lambda: au.viterbi_decoder.align_batch(lp, toks, Tl, Sl, class_mask=hint)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} T={T}: issue {(t1-t0)/200*1e6:.1f} us/call, total {(t2-t0)/200*1e6:.1f} us/call")
    t0 = time.perf_counter()
    for _ in range(100):
        fn(); torch.cuda.synchronize()
    print(f"   with a sync per call: {(time.perf_counter()-t0)/100*1e6:.1f} us/call")
