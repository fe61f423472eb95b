#!/usr/bin/env python3
"""tests/sil_time.py -- step time of the silence-anchored (segmented) mode: headline shape with SIL at ~1/12 of the
target positions and 12-40-frame planted silences (SURVEY.md section 8(d) "-sil" variant), parity sample included."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from bournemouth_forced_aligner_amd import AlignmentUtils  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
C, blank, T, S, NB = 67, 66, 1000, 40, 512
lps, toks = [], []
for _ in range(NB):
    lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=9.0, sil_rate=1 / 12, sil_len=(12, 40))
    lps.append(lp); toks.append(tk)
lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
rep = int(os.environ.get("BFA_SIL_REP", "8"))  # batch = 512 * rep
lpd = torch.from_numpy(lp).to(dev).repeat(rep, 1, 1)
tkd = torch.from_numpy(tk).repeat(rep, 1)
Tl = np.tile(T_len, rep); Sl = np.tile(S_len, rep)
au = AlignmentUtils(blank, 0)
# like bench.py: lengths and tokens device-resident, the class hint computed once on the host (no per-call H2D copies)
hint = au.viterbi_decoder.class_mask_hint(Tl, Sl, has_sil=True, n_classes=C)
tkd = tkd.to(dev).to(torch.int32)
Td = torch.from_numpy(Tl.astype(np.int32)).to(dev)
Sd = torch.from_numpy(Sl.astype(np.int32)).to(dev)
N = int(os.environ.get("BFA_SIL_STEPS", "50"))
NFL = int(os.environ.get("BFA_SIL_INFLIGHT", "1"))  # batches in flight (BatchesInFlight), 1 = plain calls
if NFL > 1:
    from bournemouth_forced_aligner_amd import BatchesInFlight
    bif = BatchesInFlight(blank, 0, n=NFL, device=dev, wait_for_caller=False)
    call = lambda: bif.submit(lpd, tkd, Td, Sd, class_mask=hint)
else:
    call = lambda: au.decode_alignments_device(lpd, tkd, Td, Sd, class_mask=hint)
for _ in range(6):
    res = call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    res = call()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
md = res.mode.cpu().numpy()
exp = ora.decode_alignments(lp[:64], tk[:64], T_len[:64], S_len[:64], ora.make_params(blank, 0), seg_cap=res.segs.shape[1])
gs, gc = res.segs[:64].cpu().numpy(), res.seg_count[:64].cpu().numpy()
mism = sum(int(gc[b] != exp["seg_count"][b] or not (gs[b, :gc[b]] == exp["seg"][b, :gc[b]]).all()) for b in range(64))
print(f"B={NB * rep} T={T} S={S} with SIL, {NFL} in flight: {ms:.3f} ms per step = {NB * rep * T / ms / 1e6:.2f} G frames/s; "
      f"segmented utterances {int((md == 1).sum())}/{len(md)}; parity sample mismatches {mism}/64")
import json  # noqa: E402
print(json.dumps({"workload": f"silence-anchored mode, phoneme head from log-probs: batch={NB * rep} T={T} S={S}, SIL at 1/12 of the "
                              f"targets, planted 12-40-frame silences; {NFL} step(s) in flight", "ms_per_step": ms,
                  "frames_per_s": NB * rep * T / (ms * 1e-3), "segmented_utterances": int((md == 1).sum()), "of": int(len(md)),
                  "algorithmic_bytes_per_frame": 4 * C + 4 * C + (4 * S + 1 + 3) // 4 + 8,
                  "hbm_frac": NB * rep * T * (4 * C + 4 * C + (4 * S + 1 + 3) // 4 + 8) / (ms * 1e-3) / 1e9 / 8000.0,
                  "parity_sample": {"utterances": 64, "mismatching_utterances": mism}}))
