#!/usr/bin/env python3
"""tools/cold_start.py -- what the FIRST call of a fresh process pays: loading libbfa_hip.so (tens of MB of gfx950 code
objects: ~40 K1 instantiations x 3 posterior-width classes), bfa_create, and the first launch of each kernel a
one-utterance call and a headline-shaped call need (the runtime loads a kernel's code object on first use).  One JSON line."""
import json
import os
import sys
import time

t_import0 = time.perf_counter()
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import synth_batch  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils, _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.lib()
t_load = time.perf_counter() - t0
t0 = time.perf_counter()
_lib.handle(0)
t_create = time.perf_counter() - t0
au = AlignmentUtils(66, 0)
lp1, tk1 = synth_batch(1, 1000, 40, 67, 1, dev)
lpB, tkB = synth_batch(512, 1000, 40, 67, 2, dev)
torch.cuda.synchronize()


def call(lp, tk):
    t = time.perf_counter()
    r = au.decode_alignments(lp, tk, [lp.shape[1]] * lp.shape[0], [tk.shape[1]] * lp.shape[0], lazy=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, r


first_one, _ = call(lp1, tk1)
second_one, _ = call(lp1, tk1)
first_batch, _ = call(lpB, tkB)
second_batch, _ = call(lpB, tkB)
print(json.dumps({"library_bytes": os.path.getsize(_lib.SO_PATH), "dlopen_ms": t_load * 1e3, "bfa_create_ms": t_create * 1e3,
                  "first_call_one_utterance_ms": first_one, "second_call_one_utterance_ms": second_one,
                  "first_call_512_utterances_ms": first_batch, "second_call_512_utterances_ms": second_batch,
                  "what": "decode_alignments(lazy=True) incl. synchronisation, device-resident posteriors, fresh process"}))
