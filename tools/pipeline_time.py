#!/usr/bin/env python3
"""tools/pipeline_time.py [B] -- where the time of PhonemeTimestampAligner.extract_timestamps_from_logits goes on a
headline-shaped batch (T=1000, S=40, both heads): device passes (HIP events) against the host-side shaping of the
result into the reference's Python tuples."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bournemouth_forced_aligner_amd import PhonemeTimestampAligner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FUSED = (sys.argv[2] != "0") if len(sys.argv) > 2 else True  # 1: bfa_align_heads from raw logits, 0: log_softmax pass + two calls
dev = torch.device("cuda", 0)
T, S = 1000, 40
lp, toks = bench.synth_batch(B, T, S, 67, 1003, dev)
g = torch.Generator(device=dev)
g.manual_seed(5)
gmap = {p: 1 + p % 15 for p in range(66)}  # no silence group among the targets, like the phoneme tokens
lut = torch.tensor([gmap[p] for p in range(66)] + [16], device=dev)
lg = torch.randn((B, T, 17), generator=g, device=dev)
lg.scatter_add_(2, lut[lp.argmax(dim=-1)].unsqueeze(-1), torch.full((B, T, 1), 7.0, device=dev))  # planted like lp
al = PhonemeTimestampAligner(device="cuda:0", phoneme_id_to_group_id=gmap)
seqs = toks.cpu().tolist()
spec = [T] * B
wl = [T * 268] * B
for as_arrays in ((True,) if len(sys.argv) > 2 else (False, "lazy", True, "tensor")):
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kw = {"as_arrays": True} if as_arrays in (True, "tensor") else ({"lazy": True} if as_arrays == "lazy" else {})
        # "tensor": the targets as the padded int tensor the reference builds at core.py:848-853 instead of lists of ids
        out = al.extract_timestamps_from_logits(lp, lg, spec, toks if as_arrays == "tensor" else seqs, wl,
                                                start_offset_times=0.0, fused=FUSED, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"B={B} fused={FUSED} as_arrays={as_arrays}: {dt * 1e3:.1f} ms per call ({B * T / dt / 1e6:.1f} M frames/s through the API)")
    import json
    print(json.dumps({"workload": f"PhonemeTimestampAligner.extract_timestamps_from_logits, batch={B} T={T} S={S}, both heads from raw "
                                  f"logits, fused={FUSED}", "result": "padded arrays" if as_arrays is True else "padded arrays, targets given as a padded tensor" if as_arrays == "tensor" else ("lazy list of dicts (lazy=True)" if as_arrays == "lazy" else "the reference's list of dicts of lists of 8-tuples"),
                      "ms_per_call_host_and_device": dt * 1e3, "frames_per_s_through_the_api": B * T / dt}))
    if hasattr(al, "last_device_ms"):
        print("   device passes:", {k: round(v, 3) for k, v in al.last_device_ms.items()})

if os.environ.get("BFA_PROFILE_HOST"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    tg = toks if os.environ["BFA_PROFILE_HOST"] == "tensor" else seqs
    for _ in range(20):
        al.extract_timestamps_from_logits(lp, lg, spec, tg, wl, start_offset_times=0.0, as_arrays=True, fused=FUSED)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
