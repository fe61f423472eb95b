#!/usr/bin/env python3
"""tools/lone_times.py -- phase times inside k_one (plan / window DP / rerun / walk) of utterance 0 of an 8-utterance call,
from a library built with -DBFA_DBG_TIMES (bournemouth-forced-aligner_amd/variants/libbfa_times.so; 10-ns ticks in mode[1..4])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bournemouth_forced_aligner_amd import AlignmentUtils
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((8, 1000, 40), (8, 600, 20), (256, 600, 20)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl = torch.full((B,), T, dtype=torch.int32, device=dev); Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    hint = au.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=67)
    for _ in range(5):
        r = au.viterbi_decoder.align_batch(lp, toks, Tl, Sl, class_mask=hint)
    torch.cuda.synchronize()
    mm = r.mode.cpu().numpy()
    m = mm[1:5] * 0.01
    print(f"B={B} T={T} S={S}: plan {m[0]:.2f} us, window DP {m[1]-m[0]:.2f} us, rerun {m[2]-m[1]:.2f} us, walk {m[3]-m[2]:.2f} us, total {m[3]:.2f} us; inside the DP loop: producer busy {mm[5]*0.01:.2f} us, consumer busy {mm[6]*0.01:.2f} us")
