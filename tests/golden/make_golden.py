#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generate the golden vectors from the REFERENCE itself.

Run in the build container (needs /root/reference; it never exists on the GPU box):
    python tests/golden/make_golden.py
Writes tests/golden/hotpath_cases.npz (inputs + the reference's outputs).  Only data is stored:
inputs (float32 log-probs, targets, flags) and what the reference returned for them.

Level 1 (forced_alignment.py / utils.py loaded by file path):
    AlignmentUtils.decode_alignments, .decode_alignments_simple,
    ViterbiDecoder.decode_with_forced_alignment (framewise), ._viterbi_decode, ._detect_silence_segments,
    ._boost_target_phonemes + ._enforce_minimum_probabilities (modified log-probs, bit pattern),
    utils._calculate_confidences, utils.convert_to_ms, F.log_softmax.
Level 2 (core.py with stub torchaudio/phonemizer, no model):
    PhonemeTimestampAligner.extract_timestamps_from_segment_batch on synthetic logits
    (ensure_target_coverage, extend_soft_boundaries_func, confidences, ms, sort) -> 8-tuples.
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


def level1(out):
    fa = refload.forced_alignment()
    ut = refload.utils()
    rng = np.random.default_rng(20260927)
    specs = []
    # (T, S, C, peak, sigma, sil_rate, anchors, truly_forced, ignore_noise, boost, enforce, repeat)
    for T, S in [(75, 7), (120, 20), (200, 33), (64, 16), (300, 9), (97, 31)]:
        for peak in (9.0, 2.0):
            specs.append((T, S, 67, peak, 1.0, 0.0, 10, True, True, True, True, 0.1))
    specs += [(150, 12, 67, 0.5, 1.0, 0.0, 10, True, True, True, True, 0.0),    # flat: sentinel regime
              (260, 40, 67, 0.3, 2.0, 0.0, 10, False, True, True, True, 0.0),
              (400, 90, 67, 1.0, 1.0, 0.0, 0, True, True, True, True, 0.2),
              (120, 20, 17, 6.0, 1.0, 0.0, 10, True, True, True, True, 0.2),    # group head
              (180, 25, 17, 1.0, 1.0, 0.2, 10, False, False, True, True, 0.0),
              (90, 12, 67, 5.0, 1.0, 0.0, 10, True, False, False, False, 0.0),  # given emissions, noise kept
              (90, 12, 67, 5.0, 1.0, 0.0, 10, True, True, False, True, 0.0),
              (90, 12, 67, 5.0, 1.0, 0.0, 10, True, True, True, False, 0.0),
              (20, 7, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),      # stride 2
              (20, 9, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),      # stride 2
              (20, 19, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),     # stride 1
              (20, 20, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),     # proportional
              (9, 10, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),      # too short -> ValueError
              (7, 1, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),
              (50, 0, 67, 5.0, 1.0, 0.0, 10, True, True, True, True, 0.0),      # empty target
              (600, 20, 67, 9.0, 1.0, 0.0, 10, True, True, True, True, 0.0),    # config C2 shape
              (1000, 40, 67, 9.0, 1.0, 0.0, 10, True, True, True, True, 0.0),   # config C3 shape
              (930, 230, 67, 4.0, 1.0, 0.0, 10, True, True, True, True, 0.05)]  # long path, L = 921
    for anchors in (10, 3, 5):
        for k in range(6):
            T = int(rng.integers(120, 420))
            S = int(rng.integers(6, T // 6))
            specs.append((T, S, 67, float(rng.choice([9.0, 5.0, 3.0])), 1.0, float(rng.choice([0.15, 0.3])), anchors,
                          bool(k % 2), True, True, True, 0.05))
    meta = []
    for i, (T, S, C, peak, sigma, sil_rate, anchors, tf, ign, boost, enf, rep) in enumerate(specs):
        blank = C - 1
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=sigma, sil_rate=sil_rate,
                                       sil_len=(4, 40), repeat_rate=rep)
        au = fa.AlignmentUtils(blank, 0, silence_anchors=anchors, ignore_noise=ign, truly_forced=tf)
        lpt = torch.from_numpy(lp)
        tkt = torch.from_numpy(tk)
        m = dict(T=T, S=S, C=C, blank=blank, sil=0, anchors=anchors, truly_forced=tf, ignore_noise=ign,
                 boost=boost, enforce=enf, error=None)
        out[f"c{i}_lp"] = lp
        out[f"c{i}_tok"] = tk.astype(np.int32)
        try:
            segs = au.decode_alignments(lpt[None], tkt[None], torch.tensor([T]), torch.tensor([S]),
                                        boost_targets=boost, enforce_minimum=enf)[0]
            out[f"c{i}_seg"] = np.array(segs, np.int32).reshape(-1, 4)
            if S > 0:
                fp, fi, score = au.viterbi_decoder.decode_with_forced_alignment(
                    lpt, tkt, return_scores=True, boost_targets=boost, enforce_minimum=enf, anchor_pauses=anchors > 0)
                out[f"c{i}_fph"] = fp.numpy().astype(np.int32)
                out[f"c{i}_fidx"] = fi.numpy().astype(np.int32)
                m["score"] = float(score)
                mod = lpt.clone()
                if boost:
                    mod = au.viterbi_decoder._boost_target_phonemes(mod, tkt)
                if enf:
                    mod = au.viterbi_decoder._enforce_minimum_probabilities(mod, tkt)
                if T <= 200:
                    out[f"c{i}_mod"] = mod.numpy()
                sil_segs = au.viterbi_decoder._detect_silence_segments(mod, sil_prob_threshold=0.9,
                                                                       min_silence_frames=max(anchors, 1))
                out[f"c{i}_sil09"] = np.array(sil_segs, np.int32).reshape(-1, 2)
            # confidences on the aligned tuples, widened like extend_soft_boundaries would (utils.py:70-113)
            fs = []
            for (ph, s, e, idx) in segs:
                fs.append((ph, max(0, s - int(rng.integers(0, 3))), min(T + 2, e + int(rng.integers(0, 5))), idx, False))
            if fs:
                cf = ut._calculate_confidences(lpt, fs)
                out[f"c{i}_conf_in"] = np.array([[f[0], f[1], f[2], f[3]] for f in fs], np.int32)
                out[f"c{i}_conf"] = np.array([c[5] for c in cf], np.float32)
                out[f"c{i}_conf_se"] = np.array([[c[1], c[2]] for c in cf], np.int32)
                ms = ut.convert_to_ms(cf, torch.tensor(T), 0.25, T * 268, 16000)
                out[f"c{i}_ms"] = np.array([[float(x[6]), float(x[7])] for x in ms], np.float32)
        except ValueError as e:
            m["error"] = str(e)
        # decode_alignments_simple on the same input (no SIL handling there)
        if S > 0 and T >= 1:
            try:
                simp = au.decode_alignments_simple(lpt[None], tkt[None], torch.tensor([T]), torch.tensor([S]))[0]
                out[f"c{i}_simple"] = np.array(simp, np.int32).reshape(-1, 4)
            except Exception as e:  # the reference raises on impossible shapes
                m["simple_error"] = type(e).__name__
        meta.append(m)
    out["meta"] = np.array(json.dumps(meta))

    # direct _viterbi_decode cases: arbitrary emissions, hand-built paths, bands, equal neighbours
    vmeta = []
    for i in range(12):
        C = 9
        T = int(rng.integers(5, 120))
        L = int(rng.integers(2, 40))
        lp = (rng.normal(0, 1.5, size=(T, C)) - (3.0 if i % 3 else 20.0)).astype(np.float32)
        path = rng.integers(0, C, size=L)
        path[0] = C - 1
        idx = np.arange(L) - 1
        bw = int(rng.choice([0, 0, 3, 8]))
        tf = bool(i % 2)
        vd = fa.ViterbiDecoder(C - 1, 0, truly_forced=tf)
        fp, fi = vd._viterbi_decode(torch.from_numpy(lp), torch.from_numpy(path), L, torch.from_numpy(idx), band_width=bw)
        out[f"v{i}_lp"] = lp
        out[f"v{i}_path"] = path.astype(np.int32)
        out[f"v{i}_fph"] = fp.numpy().astype(np.int32)
        out[f"v{i}_fidx"] = fi.numpy().astype(np.int32)
        vmeta.append(dict(T=T, L=L, C=C, bw=bw, truly_forced=tf, blank=C - 1))
    out["vmeta"] = np.array(json.dumps(vmeta))

    # F.log_softmax bit patterns (core.py:898-899)
    for C in (67, 17):
        x = rng.normal(0, 3, size=(257, C)).astype(np.float32)
        x[np.arange(257), rng.integers(0, C, 257)] += 9
        out[f"ls{C}_in"] = x
        out[f"ls{C}_out"] = torch.log_softmax(torch.from_numpy(x), dim=-1).numpy()


def level2(out):
    """core.py post-DP stages with a stubbed acoustic model."""
    al = refload.core_aligner()
    rng = np.random.default_rng(77)
    B, Tpad = 6, 220
    wav_lens, seqs, lc, lg, spec = [], [], [], [], []
    for b in range(B):
        T = int(rng.integers(90, Tpad + 1))
        S = int(rng.integers(5, T // 6))
        lp, tk, planted = cases.planted_case(rng, T, S, C=67, peak=float(rng.choice([7.0, 4.0])), sil_rate=0.15 if b % 2 else 0.0,
                                             sil_len=(10, 30))
        logits_c = np.full((Tpad, 67), 0.0, np.float32)
        logits_c[:T] = lp  # log-probs are valid logits
        logits_c[T:, 66] = 6.0
        grp = np.array(al._map_phonemes_to_groups(tk.tolist()), np.int64) if not isinstance(
            al._map_phonemes_to_groups(tk.tolist()), torch.Tensor) else al._map_phonemes_to_groups(tk.tolist()).numpy()
        pg = np.array([al.phoneme_id_to_group_id.get(int(p), 16) if int(p) != 66 else 16 for p in planted])
        logits_g = rng.normal(0, 1, size=(Tpad, 17)).astype(np.float32)
        logits_g[np.arange(T), pg] += 6.0
        logits_g[T:, 16] += 6.0
        lc.append(logits_c)
        lg.append(logits_g)
        spec.append(T)
        wav_lens.append(T * 268)
        seqs.append(tk.tolist())
    lc = torch.from_numpy(np.stack(lc))
    lg = torch.from_numpy(np.stack(lg))
    al._cupe_prediction_batch = lambda wavs, wl, ee: (lc, lg, None, list(spec))
    al.extractor = object()
    res, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), wav_lens, [list(s) for s in seqs],
                                                         start_offset_times=[0.5 * b for b in range(B)],
                                                         extract_embeddings=False, do_groups=True, debug=False)
    out["l2_logits_class"] = lc.numpy()
    out["l2_logits_group"] = lg.numpy()
    out["l2_spectral_lens"] = np.array(spec, np.int32)
    out["l2_wav_lens"] = np.array(wav_lens, np.int64)
    smax = max(len(s) for s in seqs)
    tk = np.full((B, smax), 66, np.int32)
    for b, s in enumerate(seqs):
        tk[b, :len(s)] = s
    out["l2_tokens"] = tk
    out["l2_seq_lens"] = np.array([len(s) for s in seqs], np.int32)
    for b in range(B):
        for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
            rows = res[b][key]
            out[f"l2_{short}{b}_int"] = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
            out[f"l2_{short}{b}_flt"] = np.array([[float(r[5]), float(r[6]), float(r[7])] for r in rows], np.float32).reshape(-1, 3)
    gs = []
    for s in seqs:
        g = al._map_phonemes_to_groups(s)
        gs.append(g.tolist() if isinstance(g, torch.Tensor) else list(g))
    gk = np.full((B, smax), 16, np.int32)
    for b, s in enumerate(gs):
        gk[b, :len(s)] = s
    out["l2_group_tokens"] = gk


def main():
    out = {}
    level1(out)
    level2(out)
    path = os.path.join(HERE, "hotpath_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
