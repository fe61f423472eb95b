#!/usr/bin/env python3
"""tests/golden/make_golden_minprob.py -- golden vectors for ViterbiDecoder.min_phoneme_prob != 1e-8, from the REFERENCE.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_minprob.py
Writes tests/golden/minprob_cases.npz: inputs, the float32 floor torch computed for each probability
(forced_alignment.py:70), and what the reference returned (tuples, framewise states, modified log-probs).
The reference's AlignmentUtils builds its ViterbiDecoder with the default (forced_alignment.py:850-853), so a caller
changes the floor through `alignment_utils.viterbi_decoder.min_phoneme_prob`, which is what this script does.
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
    rng = np.random.default_rng(20260929)
    out, meta = {}, []
    i = 0
    for C in (67, 17):
        for (T, S, peak, sigma, sil_rate) in [(160, 18, 1.5, 3.0, 0.0), (240, 30, 0.8, 2.5, 0.0), (300, 24, 4.0, 2.0, 0.2),
                                              (90, 30, 2.0, 3.0, 0.0)]:
            for prob in (1e-8, 1e-4, 1e-2, 0.5, 1e-20, 0.0):
                blank = C - 1
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=sigma, sil_rate=sil_rate,
                                               sil_len=(12, 30), repeat_rate=0.05)
                au = fa.AlignmentUtils(blank, 0, silence_anchors=10, ignore_noise=True, truly_forced=bool(i % 2))
                au.viterbi_decoder.min_phoneme_prob = prob
                lpt, tkt = torch.from_numpy(lp), torch.from_numpy(tk)
                segs = au.decode_alignments(lpt[None], tkt[None], torch.tensor([T]), torch.tensor([S]))[0]
                fp, fi, _ = au.viterbi_decoder.decode_with_forced_alignment(lpt, tkt)
                mod = au.viterbi_decoder._enforce_minimum_probabilities(au.viterbi_decoder._boost_target_phonemes(lpt.clone(), tkt), tkt)
                out[f"m{i}_lp"] = lp
                out[f"m{i}_tok"] = tk.astype(np.int32)
                out[f"m{i}_seg"] = np.array(segs, np.int32).reshape(-1, 4)
                out[f"m{i}_fph"] = fp.numpy().astype(np.int32)
                out[f"m{i}_fidx"] = fi.numpy().astype(np.int32)
                out[f"m{i}_mod"] = mod.numpy()
                out[f"m{i}_minlog"] = torch.log(torch.tensor(prob)).numpy().reshape(1)  # float32, forced_alignment.py:70
                meta.append(dict(T=T, S=S, C=C, blank=blank, prob=prob, truly_forced=bool(i % 2)))
                i += 1
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "minprob_cases.npz"), **out)
    print(f"wrote {i} cases")


if __name__ == "__main__":
    main()
