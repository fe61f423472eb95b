#!/usr/bin/env python3
"""tests/golden/l2_large.npz + c1_butterfly.json -- expected outputs of the REFERENCE's API-level path, produced by the
reference class itself through the stub harness of tests/refload.py (build container only; inputs and outputs are
stored, no reference code).

* l2_large.npz: 4 batches x 16 utterances through `extract_timestamps_from_segment_batch` (core.py:811-992) with the
  acoustic model stubbed by synthetic logits: both heads' 8-tuples (id, start_frame, end_frame, target_idx,
  is_estimated, confidence, start_ms, end_ms).  Batches differ in posterior sharpness, silence rate and
  `boundary_softness` (3, 3, 7, 1).  This widens the pin on ensure_target_coverage (default) / extend_soft_boundaries /
  _calculate_confidences / convert_to_ms from the 6 + 8 utterances of round 1 to 64 more.
* c1_butterfly.json: BASELINE.json configs[0] -- `process_sentence("butterfly", wav)` (core.py:1553) on a synthetic
  75-frame posterior with the ph66 ids [29,10,58,9,43,56,23] of examples/samples/audio/109867__timkahn__butterfly.vs.json,
  phonemiser and model stubbed: the complete result dict (coverage_analysis, phoneme_ts, group_ts, words_ts).
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


def make_batch(al, rng, B, Tpad, peak_choices, sil_rate, flat_every=0):
    wav_lens, seqs, lc, lg, spec = [], [], [], [], []
    for b in range(B):
        T = int(rng.integers(60, Tpad + 1))
        S = int(rng.integers(3, max(4, T // 6)))
        peak = float(rng.choice(peak_choices))
        if flat_every and b % flat_every == flat_every - 1:
            peak = 1.5  # flat posteriors: low confidences, long soft-boundary walks
        lp, tk, planted = cases.planted_case(rng, T, S, C=67, peak=peak, sil_rate=sil_rate if b % 2 else 0.0,
                                             sil_len=(10, 30), repeat_rate=0.05)
        logits_c = np.zeros((Tpad, 67), np.float32)
        logits_c[:T] = lp * np.float32(rng.choice([1.0, 1.0, 0.7]))  # (log-probs are valid logits; 0.7 flattens them)
        logits_c[T:, 66] = 6.0
        pg = np.array([al.phoneme_id_to_group_id.get(int(p), 16) if int(p) != 66 else 16 for p in planted])
        logits_g = rng.normal(0, 1, size=(Tpad, 17)).astype(np.float32)
        logits_g[np.arange(T), pg] += np.float32(peak * 0.8)
        logits_g[T:, 16] += 6.0
        lc.append(logits_c)
        lg.append(logits_g)
        spec.append(T)
        wav_lens.append(T * 268 + int(rng.integers(0, 268)))
        seqs.append([int(x) for x in tk])
    return torch.from_numpy(np.stack(lc)), torch.from_numpy(np.stack(lg)), spec, wav_lens, seqs


def main():
    out = {}
    rng = np.random.default_rng(20260928)
    configs = [dict(peaks=[8.0, 6.0], sil=0.0, soft=3, flat=0), dict(peaks=[7.0, 4.0], sil=0.2, soft=3, flat=4),
               dict(peaks=[6.0, 3.0], sil=0.15, soft=7, flat=5), dict(peaks=[5.0, 2.5], sil=0.1, soft=1, flat=3)]
    n_rows = 0
    for k, cfg in enumerate(configs):
        al = refload.core_aligner(boundary_softness=cfg["soft"])
        al.warn_level = 0
        B, Tpad = 16, 400
        lc, lg, spec, wav_lens, seqs = make_batch(al, rng, B, Tpad, cfg["peaks"], cfg["sil"], cfg["flat"])
        al._cupe_prediction_batch = lambda wavs, wl, ee, _r=(lc, lg, spec): (_r[0], _r[1], None, list(_r[2]))
        al.extractor = object()
        offs = [round(0.25 * b, 2) for b in range(B)]
        res, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), wav_lens, [list(s) for s in seqs],
                                                             start_offset_times=offs, extract_embeddings=False,
                                                             do_groups=True, debug=False)
        pre = f"b{k}_"
        out[pre + "logits_class"], out[pre + "logits_group"] = lc.numpy(), lg.numpy()
        out[pre + "spectral_lens"] = np.array(spec, np.int32)
        out[pre + "wav_lens"] = np.array(wav_lens, np.int64)
        out[pre + "offsets"] = np.array(offs, np.float64)
        out[pre + "softness"] = np.array(cfg["soft"], np.int32)
        smax = max(len(s) for s in seqs)
        tk = np.full((B, smax), 66, np.int32)
        gk = np.full((B, smax), 16, np.int32)
        for b, s in enumerate(seqs):
            tk[b, :len(s)] = s
            g = al._map_phonemes_to_groups(s)
            gk[b, :len(s)] = g.tolist() if isinstance(g, torch.Tensor) else list(g)
        out[pre + "tokens"], out[pre + "group_tokens"] = tk, gk
        out[pre + "seq_lens"] = np.array([len(s) for s in seqs], np.int32)
        for b in range(B):
            for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
                rows = res[b][key]
                n_rows += len(rows)
                out[f"{pre}{short}{b}_int"] = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
                out[f"{pre}{short}{b}_flt"] = np.array([[float(r[5]), float(r[6]), float(r[7])] for r in rows], np.float32).reshape(-1, 3)
    out["n_batches"] = np.array(len(configs), np.int32)
    path = os.path.join(HERE, "l2_large.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", n_rows, "reference tuples")

    # ---- C1: process_sentence("butterfly")
    al = refload.core_aligner()
    al.warn_level = 0
    toks = [29, 10, 58, 9, 43, 56, 23]
    T = 75
    r2 = np.random.default_rng(1)
    planted = np.full(T, 66)
    for j, t in enumerate(toks):
        planted[5 + 9 * j: 5 + 9 * j + 6] = t
    logits = r2.normal(0, 1, (1, T, 67)).astype(np.float32)
    logits[0, np.arange(T), planted] += 8
    groups = [int(al.phoneme_id_to_group_id.get(t, 16)) for t in toks]
    pgl = np.array([al.phoneme_id_to_group_id.get(int(p), 16) if int(p) != 66 else 16 for p in planted])
    lgrp = r2.normal(0, 1, (1, T, 17)).astype(np.float32)
    lgrp[0, np.arange(T), pgl] += 6
    ts = {"ph66": list(toks), "pg16": list(groups), "eipa": ["b", "ʌ", "ɾ", "ɚ", "f", "l", "aɪ"], "words": ["butterfly"],
          "word_num": [0] * 7}
    al.phonemize_sentence = lambda text: dict(ts, **{al.phonemes_key: list(toks), al.phoneme_groups_key: list(groups)})
    al._cupe_prediction_batch = lambda wavs, wl, ee: (torch.from_numpy(logits), torch.from_numpy(lgrp), None, [T])
    al.extractor = object()
    wav = torch.from_numpy(r2.normal(0, 0.1, (1, T * 268)).astype(np.float32))
    res = al.process_sentence("butterfly", wav, do_groups=True)
    seg = res["segments"][0]
    labels_p = {str(int(p["phoneme_id"])): p["phoneme_label"] for p in seg["phoneme_ts"]}
    labels_g = {str(int(g["group_id"])): g["group_label"] for g in seg["group_ts"]}
    json.dump({"tokens": toks, "groups": groups, "T": T, "ts": ts, "wav_samples": T * 268,
               "phoneme_labels": labels_p, "group_labels": labels_g, "expected": res},
              open(os.path.join(HERE, "c1_butterfly.json"), "w"), ensure_ascii=False, indent=1)
    np.savez_compressed(os.path.join(HERE, "c1_butterfly.npz"), logits_class=logits, logits_group=lgrp)
    print("wrote c1_butterfly.json / .npz:", len(seg["phoneme_ts"]), "phonemes,", len(seg.get("words_ts", [])), "words")


if __name__ == "__main__":
    main()
