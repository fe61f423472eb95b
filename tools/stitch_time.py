#!/usr/bin/env python3
"""tools/stitch_time.py -- throughput of bfa_stitch_windows on a headline-sized batch (B=4096, ~1000 frames, C=67)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bournemouth_forced_aligner_amd import stich_window_predictions, stitch_total_frames  # noqa: E402

dev = torch.device("cuda", 0)
B, NW, F, C = 4096, 199, 10, 67
alen = 16000 * (160 + 80 * (NW - 1)) // 1000
x = torch.randn((B, NW, F, C), device=dev)
total = stitch_total_frames(alen, F)
for pad in (None, 68):
    for _ in range(2):
        y = stich_window_predictions(x, alen, F, row_stride=pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = stich_window_predictions(x, alen, F, row_stride=pad)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = (x.numel() + B * total * C) * 4 / 1e9
    print(f"row_stride={pad}: {ms:.3f} ms for {gb:.2f} GB read+written = {gb / ms:.2f} TB/s ({total} frames)")
