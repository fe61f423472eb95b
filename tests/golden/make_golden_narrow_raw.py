#!/usr/bin/env python3
"""tests/golden/make_golden_narrow_raw.py -- RAW-LOGIT input at posterior widths below one vector (C < 16), from the REFERENCE.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_narrow_raw.py
core.py:898-899 applies F.log_softmax to the model's logits before anything else; bfa_align_heads fuses that into the
alignment and hands the later stages (logits, row statistics) instead of log-probs.  Below sixteen columns torch sums the
exponentials one after the other, so the fused front end, the on-demand row statistics of the sparse readers and the
statistics the alignment writes all have to follow that order.  Writes tests/golden/narrow_raw_cases.npz: logits, targets,
and what the reference returns from log_softmax -> decode_alignments (tuples, framewise states), _calculate_confidences of
those tuples (utils.py:70-113) and extend_soft_boundaries_func (core.py:682-809).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import refload  # noqa: E402

torch.set_num_threads(1)


def main():
    fa, ut = refload.forced_alignment(), refload.utils()
    al = refload.core_aligner()
    rng = np.random.default_rng(20261001)
    out, meta = {}, []
    i = 0
    for C in (2, 3, 4, 5, 8, 11, 12, 15):
        for (T, S, peak, sigma) in [(70, 9, 4.0, 1.0), (120, 25, 1.0, 2.0), (64, 16, 0.3, 1.0), (33, 30, 3.0, 1.0)]:
            blank = C - 1
            if C == 2:
                S = min(S, 6)
            lp, tk, _ = cases.planted_case(rng, T, S, C=max(C, 3), blank=max(C, 3) - 1, sil=0, peak=peak, sigma=sigma, repeat_rate=0.1)
            if C <= 3:  # (ids 0 = SIL and C-1 = blank: C = 3 leaves one phoneme, C = 2 only SIL)
                tk = np.ones_like(tk) if C == 3 else np.zeros_like(tk)
                lp = lp[:, :C]
            # raw logits: any real matrix; scaled log-probs plus a row offset, so that log_softmax has work to do
            x = (lp * np.float32(1.7) + rng.normal(0.0, 3.0, size=(T, 1)).astype(np.float32)).astype(np.float32)
            au = fa.AlignmentUtils(blank, 0, silence_anchors=0, ignore_noise=True, truly_forced=bool(i % 2))
            xt, tkt = torch.from_numpy(x), torch.from_numpy(tk)
            lpr = torch.log_softmax(xt[None], dim=2)  # core.py:898
            segs = au.decode_alignments(lpr, tkt[None], torch.tensor([T]), torch.tensor([S]))[0]
            fp, fi, _ = au.viterbi_decoder.decode_with_forced_alignment(lpr[0], tkt, anchor_pauses=False)  # (as decode_alignments calls it with silence_anchors = 0)
            five = [(p, s, e, ix, False) for (p, s, e, ix) in segs]
            conf = [float(r[5]) for r in ut._calculate_confidences(lpr[0], five)]
            ext = al.extend_soft_boundaries_func(lpr, [five], boundary_softness=3)[0]
            out[f"r{i}_x"] = x
            out[f"r{i}_tok"] = tk.astype(np.int32)
            out[f"r{i}_seg"] = np.array(segs, np.int32).reshape(-1, 4)
            out[f"r{i}_fph"] = fp.numpy().astype(np.int32)
            out[f"r{i}_fidx"] = fi.numpy().astype(np.int32)
            out[f"r{i}_conf"] = np.array(conf, np.float32)
            out[f"r{i}_ext"] = np.array([[r[0], r[1], r[2], r[3]] for r in ext], np.int32).reshape(-1, 4)
            meta.append(dict(T=T, S=int(S), C=C, blank=blank, truly_forced=bool(i % 2)))
            i += 1
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "narrow_raw_cases.npz"), **out)
    print(f"wrote {i} cases")


if __name__ == "__main__":
    main()
