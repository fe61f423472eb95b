#!/usr/bin/env python3
"""tools/api_time.py -- what the drop-in entry points deliver on the headline batch (B=4096, T=1000, S=40): host + device
time of AlignmentUtils.decode_alignments (lists of tuples, forced_alignment.py:856-910) against the device-resident call,
and of AlignmentResult.to_lists() alone.  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402

dev = torch.device("cuda", 0)
B, T, S = 4096, 1000, 40
lp, toks = bench.synth_batch(B, T, S, 67, 1003, dev)
Tl = torch.full((B,), T, dtype=torch.int32, device=dev)
Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
au = AlignmentUtils(66, 0)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r


dev_ms, res = timed(lambda: au.decode_alignments_device(lp, toks, Tl, Sl))
lists_ms, _ = timed(lambda: res.to_lists())                       # the lazy list of lists (one packed copy to the host)
touch1_ms, _ = timed(lambda: res.to_lists()[17])                  # ... and one utterance's tuples
all_ms, _ = timed(lambda: res.to_lists().tolist())                # ... and every tuple (one numpy pass)
iter_ms, _ = timed(lambda: sum(len(r) for r in res.to_lists()))   # ... every tuple through iteration
full_ms, rows = timed(lambda: au.decode_alignments(lp, toks, Tl, Sl))              # the reference's return type: list of lists of tuples
lazy_ms, _ = timed(lambda: au.decode_alignments(lp, toks, Tl, Sl, lazy=True))       # LazyRowLists (opt-in)
assert type(rows) is list and type(rows[0]) is list and type(rows[0][0]) is tuple
print(json.dumps({"workload": f"batch={B} T={T} S={S} ph66, device-resident inputs", "decode_alignments_device_ms": dev_ms,
                  "to_lists_ms": lists_ms, "to_lists_one_utterance_ms": touch1_ms, "to_lists_every_tuple_ms": all_ms,
                  "to_lists_iterate_ms": iter_ms, "decode_alignments_ms": full_ms, "decode_alignments_lazy_ms": lazy_ms, "tuples": sum(len(r) for r in rows),
                  "frames_per_s_through_decode_alignments": B * T / (full_ms * 1e-3)}))
