#!/usr/bin/env python3
"""tools/mix_class.py -- ONE window class at a time through the three bodies that can align it (where do
k_mix's instructions go?).  B utterances of T frames and S targets (S picks the class: 20 -> Rw 1, 40 -> 2, 80 -> 3, 115 -> 4),
standard mode, reference-default flags:
  --mode fast   uniform-length hint, window routing off: the fast window's class kernel (k_dp4w<Rw,..>)
  --mode exact  uniform-length hint, routing always: the exact window's class kernel (k_dp4x_redo<Rw,..>)
  --mode mix    no uniform hint: k_mix (exact window, DP + walk of an utterance in one workgroup)
Prints one JSON line; run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU` (tools/r6_mixclass.sh) for instructions per frame."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import synth_batch  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils, _lib  # noqa: E402

ap_ = argparse.ArgumentParser()
ap_.add_argument("--batch", type=int, default=4096)
ap_.add_argument("--frames", type=int, default=1000)
ap_.add_argument("--tokens", type=int, default=40)
ap_.add_argument("--mode", choices=("fast", "exact", "mix"), default="mix")
ap_.add_argument("--calls", type=int, default=6)
args = ap_.parse_args()
dev = torch.device("cuda", 0)
B, T, S, C = args.batch, args.frames, args.tokens, 67
lp, tk = synth_batch(B, T, S, C, 1003, dev, peak=9.0)
Tl, Sl = np.full(B, T, np.int64), np.full(B, S, np.int64)
Td, Sd = torch.from_numpy(Tl.astype(np.int32)).to(dev), torch.from_numpy(Sl.astype(np.int32)).to(dev)
au = AlignmentUtils(blank_id=C - 1, silence_id=0)
if S > 64:
    au.viterbi_decoder.window_max_tokens = 100000   # (the fast window is not tried on more than 64 targets by default)
vd = au.viterbi_decoder
hint, path = vd.hint_and_path(Tl, Sl, False, n_classes=C, Smax=tk.shape[1])
if args.mode == "mix":
    hint &= ~_lib.HINT_UNIFORM_LENGTHS
_lib.handle(0, 0)
_lib.set_window_routing(0, 0, {"fast": 0, "exact": 2, "mix": 1}[args.mode])
fn = lambda: au.decode_alignments_device(lp, tk, Td, Sd, class_mask=hint)  # noqa: E731
for _ in range(3):
    r = fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.calls):
    r = fn()
e1.record()
torch.cuda.synchronize()
cnt = r.call_counters()
print(json.dumps({"mode": args.mode, "B": B, "T": T, "S": S, "frames": B * T, "ms_per_call": e0.elapsed_time(e1) / args.calls,
                  "status_ok": bool((r.status.cpu() == 0).all()), "counters": cnt}), flush=True)
