#!/usr/bin/env python3
"""tools/one_long.py [B] [T] [S] -- repeated alignment of one shape (for kernel traces of small / long batches)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 120
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
for _ in range(30):
    au.viterbi_decoder.align_batch(lp, toks, [T] * B, [S] * B)
torch.cuda.synchronize()
