#!/usr/bin/env python3
"""tools/lsm_time.py -- throughput of bfa_log_softmax (core.py:898-899 on the device) on headline-sized logits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bournemouth_forced_aligner_amd import log_softmax  # noqa: E402

dev = torch.device("cuda", 0)
for C in (67, 17):
    x = torch.randn((4096, 1000, C), device=dev)
    for _ in range(2):
        y = log_softmax(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = log_softmax(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = 2 * x.numel() * 4 / 1e9
    print(f"C={C}: {ms:.3f} ms for {gb:.2f} GB read+written = {gb / ms:.2f} TB/s")
