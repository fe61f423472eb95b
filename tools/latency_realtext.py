#!/usr/bin/env python3
"""tools/latency_realtext.py -- the reference's own call shape: `process_segments` feeds the path in
chunks of batch_size = 16 (core.py:1212,1348-1389), both heads, SIL in the targets, segments of at most 30 s (T <= 1 870).

For B in {1, 4, 16, 64}: the device time of one bfa_align_heads call with the post-DP stages (HIP events around it on the
caller's stream, calls back to back) and the host + device time of PhonemeTimestampAligner.extract_timestamps_from_logits with
the reference's default result type and with `as_arrays`.  One JSON line per B."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import group_lut, synth_realtext_ragged  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils, PhonemeTimestampAligner  # noqa: E402
from bournemouth_forced_aligner_amd.forced_alignment import align_heads  # noqa: E402

dev = torch.device("cuda", 0)
peak = float(os.environ.get("BFA_PEAK", "9"))
tok_div = int(os.environ.get("BFA_TOK_DIV", "12"))
lut = group_lut()
gmap = {p: int(lut[p]) for p in range(66)}
al = PhonemeTimestampAligner(device="cuda:0", phoneme_id_to_group_id=gmap)
ap, ag = AlignmentUtils(66, 0), AlignmentUtils(16, 0)
BS = [int(v) for v in os.environ.get("BFA_BS", "1,4,16,64").split(",")]
DEVICE_ONLY = bool(int(os.environ.get("BFA_DEVICE_ONLY", "0")))   # (kernel timelines: a few alignment calls, nothing else)
if os.environ.get("BFA_WIDE_ANY_MAX"):   # (A/B of BFA_OPT_WIDE_ANY_MAX_BATCH, tools/r6_thresh.sh)
    from bournemouth_forced_aligner_amd import _lib
    _lib.check(_lib.lib().bfa_set_option(_lib.handle(0, 0), _lib.OPT_WIDE_ANY_MAX_BATCH, int(os.environ["BFA_WIDE_ANY_MAX"])), _lib.handle(0, 0), "bfa_set_option")
for B in BS:
    xp, xg, tp, tg, Tl, Sl = synth_realtext_ragged(B, 300, 1870, tok_div, 3000 + B, dev, peak=peak, gpeak=max(1.0, peak - 2.0))
    Tn, Sn = Tl.numpy().astype(np.int64), Sl.numpy().astype(np.int64)
    Td, Sd = Tl.to(dev), Sl.to(dev)
    vd = ap.viterbi_decoder
    hints = [vd.class_mask_hint(Tn, Sn, has_sil=True, n_classes=67), vd.class_mask_hint(Tn, Sn, has_sil=True, n_classes=17)]
    fn = lambda: align_heads([ap, ag], [xp, xg], [tp, tg], Td, Sd, class_masks=hints, post={"extend": True, "boundary_softness": 3})  # noqa: E731
    for _ in range(5):
        r = fn()
    torch.cuda.synchronize()
    assert all(int((x[0].status != 0).sum()) == 0 for x in r)
    if DEVICE_ONLY:
        continue
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / n
    # one call at a time with a synchronisation behind each (what a caller that needs the result sees)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    sync_ms = (time.perf_counter() - t0) / n * 1e3
    seqs = [tp[b, :int(Sn[b])].cpu().tolist() for b in range(B)]
    spec = [int(t) for t in Tn]
    wl = [int(t) * 256 for t in Tn]
    host = {}
    for name, kw in (("default", {}), ("as_arrays", {"as_arrays": True})):
        for _ in range(3):
            al.extract_timestamps_from_logits(xp, xg, spec, seqs, wl, start_offset_times=0.0, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            al.extract_timestamps_from_logits(xp, xg, spec, seqs, wl, start_offset_times=0.0, **kw)
        host[name] = (time.perf_counter() - t0) / 20 * 1e3
    modes = [int((x[0].mode == 1).sum()) for x in r]
    print(json.dumps({"B": B, "frames": int(Tn.sum()), "longest_T": int(Tn.max()), "longest_S": int(Sn.max()), "peak": peak,
                      "tok_div": tok_div, "device_ms_back_to_back": dev_ms, "ms_per_call_with_sync": sync_ms,
                      "extract_timestamps_from_logits_ms": host, "segmented_utterances": modes,
                      "what": "bfa_align_heads + post-DP stages of both heads (device, HIP events over 50 calls); "
                              "extract_timestamps_from_logits incl. host shaping"}), flush=True)
