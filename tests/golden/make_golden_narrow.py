#!/usr/bin/env python3
"""tests/golden/make_golden_narrow.py -- golden vectors for posterior widths below one vector (C < 16), from the REFERENCE.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_narrow.py
The reference boosts and re-normalises rows of any width (forced_alignment.py:29-56); torch's CPU log_softmax sums fewer
than sixteen exponentials one after the other instead of in sixteen lane accumulators, so these widths pin a different
summation order.  Writes tests/golden/narrow_cases.npz: inputs, F.log_softmax of random rows (bit patterns), the
reference's boosted / floored emissions, framewise states and tuples for C in {2..15}.
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
    fa = refload.forced_alignment()
    rng = np.random.default_rng(20260930)
    out, meta = {}, []
    i = 0
    for C in (3, 4, 5, 8, 11, 12, 15):
        rows = (rng.normal(0.0, 4.0, size=(257, C)) * rng.choice([0.1, 1.0, 6.0], size=(257, 1))).astype(np.float32)
        out[f"ls{C}_x"] = rows
        out[f"ls{C}_y"] = torch.log_softmax(torch.from_numpy(rows), dim=-1).numpy()
        for (T, S, peak, sigma) in [(70, 9, 4.0, 1.0), (120, 25, 1.0, 2.0), (64, 16, 0.3, 1.0), (31, 30, 3.0, 1.0)]:
            blank = C - 1
            lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, sil=0, peak=peak, sigma=sigma, repeat_rate=0.1)
            if C == 3:  # (ids 0 = SIL and 2 = blank leave one phoneme)
                tk = np.ones_like(tk)
            au = fa.AlignmentUtils(blank, 0, silence_anchors=0, ignore_noise=True, truly_forced=bool(i % 2))
            lpt, tkt = torch.from_numpy(lp), torch.from_numpy(tk)
            segs = au.decode_alignments(lpt[None], tkt[None], torch.tensor([T]), torch.tensor([S]))[0]
            fp, fi, _ = au.viterbi_decoder.decode_with_forced_alignment(lpt, tkt)
            mod = au.viterbi_decoder._enforce_minimum_probabilities(au.viterbi_decoder._boost_target_phonemes(lpt.clone(), tkt), tkt)
            out[f"n{i}_lp"] = lp
            out[f"n{i}_tok"] = tk.astype(np.int32)
            out[f"n{i}_seg"] = np.array(segs, np.int32).reshape(-1, 4)
            out[f"n{i}_fph"] = fp.numpy().astype(np.int32)
            out[f"n{i}_fidx"] = fi.numpy().astype(np.int32)
            out[f"n{i}_mod"] = mod.numpy()
            meta.append(dict(T=T, S=S, C=C, blank=blank, truly_forced=bool(i % 2)))
            i += 1
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "narrow_cases.npz"), **out)
    print(f"wrote {i} cases")


if __name__ == "__main__":
    main()
