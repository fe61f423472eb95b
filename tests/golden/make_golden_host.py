#!/usr/bin/env python3
"""tests/golden/host_cases.json: expected outputs of the reference's host-side result shaping
(core.py: _align_words :1062, analyze_alignment_coverage :1701, compress_frames :1783, framewise_assortment
:1813, post_process_segment :1140) and of extract_timestamps_from_segment_simplified (:995) on the Level-2
logits already held in hotpath_cases.npz.  Produced by the reference class itself, constructed through the stub
harness in tests/refload.py.  Run in the build container (the reference never ships to the GPU box)."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refload  # noqa: E402


def rand_phoneme_ts(rng, n, t0=0.0, gap_p=0.3):
    rows, t = [], t0
    for i in range(n):
        if rng.random() < gap_p:
            t += float(rng.uniform(1, 400))
        d = float(rng.uniform(5, 180))
        rows.append({"phoneme_id": int(rng.integers(0, 66)), "ipa_label": f"p{int(rng.integers(0, 30))}",
                     "start_ms": t, "end_ms": t + d, "confidence": float(rng.uniform(0, 1)), "index": i,
                     "group_id": int(rng.integers(0, 16))})
        t += d - (float(rng.uniform(0, 20)) if rng.random() < 0.2 else 0.0)  # some overlap
    return rows


def main():
    al = refload.core_aligner()
    al.warn_level = 0
    rng = np.random.default_rng(20260927)
    out = {"words": [], "coverage": [], "compress": [], "framewise": [], "post": [], "simplified": []}

    # _align_words: equal lengths, shorter / longer word_num, one-phoneme last word, unknown word index, empties
    for case in range(40):
        n = int(rng.integers(0, 25))
        pts = rand_phoneme_ts(rng, n)
        nw = int(rng.integers(1, 7))
        m = max(0, n + int(rng.integers(-3, 4))) if case % 3 else n
        wn = sorted(int(rng.integers(0, nw + (case % 5 == 0))) for _ in range(m))
        if case % 7 == 0 and m >= 2:
            wn[-1] = wn[-2] + 1  # final one-phoneme word
        words = [f"w{i}" for i in range(nw)]
        exp = al._align_words(copy.deepcopy(pts), list(wn), list(words))
        out["words"].append({"phoneme_ts": pts, "word_num": wn, "words": words, "expected": exp})

    labels = {i: f"L{i}" for i in range(0, 66, 2)}
    for case in range(12):
        tgt = [int(x) for x in rng.integers(0, 20, int(rng.integers(0, 15)))]
        ali = [(int(x), 0, 1, 0) for x in rng.integers(0, 24, int(rng.integers(0, 15)))]
        exp = al.analyze_alignment_coverage(torch.tensor(tgt) if case % 2 else tgt, ali, labels)
        exp["missing_phonemes"] = sorted(exp["missing_phonemes"])  # set iteration order is not part of the contract
        exp["extra_phonemes"] = sorted(exp["extra_phonemes"])
        out["coverage"].append({"target": tgt, "aligned": ali, "tensor": bool(case % 2), "expected": exp})

    for case in range(8):
        fr = [int(x) for x in np.repeat(rng.integers(0, 5, 12), rng.integers(1, 6, 12))][: int(rng.integers(0, 40))]
        exp = al.compress_frames(list(fr))
        assert al.decompress_frames(exp) == fr
        out["compress"].append({"frames": fr, "expected": [list(x) for x in exp]})

    for case in range(60):
        n = int(rng.integers(0, 20))
        off = float(rng.choice([0.0, 500.0, 1234.5]))
        pts = rand_phoneme_ts(rng, n, t0=off + float(rng.uniform(-50, 300)), gap_p=0.5)
        if case % 6 == 0:
            rng.shuffle(pts)  # the function sorts in place
        fps = float(rng.choice([86.1328125, 100.0, 50.0, 93.75]))
        total = int(rng.integers(1, 400))
        gc = int(rng.choice([5, 5, 0, 1, 3, 12]))
        key = ["phoneme_id", "ipa_label", "group_id"][case % 3]
        exp = al.framewise_assortment(copy.deepcopy(pts), total, fps, gap_contraction=gc, select_key=key, offset_ms=off)
        out["framewise"].append({"ts": pts, "total_frames": total, "fps": fps, "gap_contraction": gc, "key": key,
                                 "offset_ms": off, "expected": exp})

    # post_process_segment on Level-2 style rows
    class _Ph:
        index_to_plabel = {i: f"P{i}" for i in range(60)}
        index_to_glabel = {i: f"G{i}" for i in range(14)}
    al.phonemizer = _Ph()
    for case in range(6):
        S = int(rng.integers(1, 14))
        seq = [int(x) for x in rng.integers(0, 66, S)]
        rows, t = [], 0.0
        for j in range(S):
            if rng.random() < 0.15:
                continue
            d = float(rng.uniform(10, 90))
            rows.append((seq[j], int(t / 16), int((t + d) / 16), j, bool(rng.random() < 0.1), float(rng.uniform(0, 1)), t, t + d))
            t += d
        grows = [(int(r[0]) % 17, r[1], r[2], r[3], r[4], r[5], r[6], r[7]) for r in rows]
        nw = int(rng.integers(1, 4))
        ts = {"eipa": [f"e{j}" for j in range(S - (case == 3))], "word_num": sorted(int(rng.integers(0, nw)) for _ in range(S)),
              "words": [f"w{i}" for i in range(nw)], "ph66": seq}
        seg = {"start": 0.25, "end": 1.5, "text": "t"}
        exp = al.post_process_segment(dict(seg), dict(ts), torch.tensor(seq), list(rows), list(grows) if case % 2 else None)
        exp["coverage_analysis"]["missing_phonemes"] = sorted(exp["coverage_analysis"]["missing_phonemes"])
        exp["coverage_analysis"]["extra_phonemes"] = sorted(exp["coverage_analysis"]["extra_phonemes"])
        out["post"].append({"segment": seg, "ts": ts, "seq": seq, "rows": [list(r) for r in rows],
                            "grows": [list(r) for r in grows] if case % 2 else None, "expected": exp})

    # extract_timestamps_from_segment_simplified on the Level-2 logits of hotpath_cases.npz
    gold = np.load(os.path.join(HERE, "hotpath_cases.npz"))
    lc = torch.from_numpy(gold["l2_logits_class"])
    spec = gold["l2_spectral_lens"].tolist()
    B = lc.shape[0]
    seqs = [gold["l2_tokens"][b, :gold["l2_seq_lens"][b]].tolist() for b in range(B)]
    al._cupe_prediction_batch = lambda wavs, wl, extract_embeddings=False: (lc, None, None, list(spec))
    res = al.extract_timestamps_from_segment_simplified(torch.zeros(B, 16), gold["l2_wav_lens"].tolist(), [list(s) for s in seqs],
                                                        start_offset_times=[0.5 * b for b in range(B)], debug=False)
    for b in range(B):
        out["simplified"].append([[int(r[0]), int(r[1]), int(r[2]), int(r[3]), bool(r[4]), float(r[5]), float(r[6]), float(r[7])]
                                  for r in res[b]["phoneme_timestamps"]])
    # ensure_target_coverage with ensure_completeness=True (core.py:462-679) on synthetic aligner outputs
    alc = refload.core_aligner(ensure_completeness=True)
    alc.warn_level = 0
    alc.phonemizer = _Ph()
    out["complete"] = []
    for case in range(300):
        S = int(rng.integers(1, 16))
        seq = [int(x) for x in rng.choice([0, 0, 3, 7, 11, 20, 41], S)]
        if case % 4 == 0:
            k = int(rng.integers(1, min(S, 3) + 1))
            seq[-k:] = [0] * k  # trailing silences
        rows, t = [], int(rng.integers(0, 6))
        keep_p = float(rng.choice([1.0, 0.8, 0.5, 0.2]))
        last_aligned = S if case % 3 else int(rng.integers(0, S + 1))  # targets past it are missing (trailing)
        for j in range(S):
            d = int(rng.integers(1, 9))
            if j < last_aligned and rng.random() < keep_p:
                rows.append((seq[j], t, t + d, j))
                if rng.random() < 0.2:  # the same target again: touching, overlapping or apart
                    t2 = t + d + int(rng.integers(-2, 4))
                    d2 = int(rng.integers(1, 9))
                    rows.append((seq[j], t2, t2 + d2, j))
                    t = max(t, t2 + d2 - d)
                t += d
            elif rng.random() < 0.5:
                t += d
        if case % 5 == 0 and rows:
            rows.insert(int(rng.integers(0, len(rows) + 1)), (9, 2, 4, -1))
        if case % 11 == 0:
            rows.append((9, t, t + 2, S + int(rng.integers(0, 3))))
        pad = int(rng.integers(0, 3))
        entry = {"seq": seq + [66] * pad, "len": S, "rows": [list(r) for r in rows], "sil": 0}
        try:
            got = alc.ensure_target_coverage([torch.tensor(entry["seq"])], [[tuple(r) for r in rows]], seq_lens=[S],
                                             _silence_class=0)
            entry["expected"] = [[int(r[0]), int(r[1]), int(r[2]), int(r[3]), bool(r[4])] for r in got[0]]
        except Exception as e:  # the reference's own consistency check (:676-677)
            if not str(e).startswith("Post-processing error"):
                raise
            entry["expected"] = "raises"
        out["complete"].append(entry)
    print("complete: raising cases", sum(1 for c in out["complete"] if c["expected"] == "raises"),
          "estimated rows", sum(sum(1 for r in c["expected"] if r[4]) for c in out["complete"] if c["expected"] != "raises"))

    # the whole Level-2 pipeline with ensure_completeness=True on utterances where stride 1 lets the path skip
    # targets (T barely above S, posteriors planted on a subset of the targets) -> l2_complete.npz
    Bc, Tpad = 8, 64
    lcs, lgs, specs, wls, seqs_c = [], [], [], [], []
    for b in range(Bc):
        T = int(rng.integers(24, Tpad + 1))
        S = int(rng.integers(max(4, T // 2 + 1), T - 1))  # stride 1: S + 1 <= T < 2S + 1
        tk = []
        while len(tk) < S:
            x = int(rng.integers(0 if b % 2 else 1, 40))
            if not tk or tk[-1] != x:
                tk.append(x)
        if b % 3 == 0:
            tk[-2:] = [5, 0] if tk[-3] != 5 else [6, 0]  # a trailing silence target
        keep = sorted(rng.choice(S, size=max(2, int(S * 0.6)), replace=False).tolist())
        planted = np.full(T, 66)
        bounds = np.linspace(0, T, len(keep) + 1).astype(int)
        for k, j in enumerate(keep):
            planted[bounds[k]: max(bounds[k] + 1, bounds[k + 1] - (1 if k % 2 else 0))] = tk[j]
        lc1 = rng.normal(0, 1, (Tpad, 67)).astype(np.float32)
        lc1[np.arange(T), planted] += 9.0
        lc1[T:, 66] += 6.0
        gseq = alc._map_phonemes_to_groups(tk)
        gseq = gseq.tolist() if isinstance(gseq, torch.Tensor) else list(gseq)
        pg = np.array([alc.phoneme_id_to_group_id.get(int(p), 16) if int(p) != 66 else 16 for p in planted])
        lg1 = rng.normal(0, 1, (Tpad, 17)).astype(np.float32)
        lg1[np.arange(T), pg] += 7.0
        lg1[T:, 16] += 6.0
        lcs.append(lc1); lgs.append(lg1); specs.append(T); wls.append(T * 268); seqs_c.append(tk)
    lc2 = torch.from_numpy(np.stack(lcs))
    lg2 = torch.from_numpy(np.stack(lgs))
    alc._cupe_prediction_batch = lambda wavs, wl, ee: (lc2, lg2, None, list(specs))
    alc.extractor = object()
    resc, _, _ = alc.extract_timestamps_from_segment_batch(torch.zeros(Bc, 16), wls, [list(s) for s in seqs_c],
                                                           start_offset_times=0.25, extract_embeddings=False,
                                                           do_groups=True, debug=False)
    z = {"logits_class": lc2.numpy(), "logits_group": lg2.numpy(), "spectral_lens": np.array(specs, np.int32),
         "wav_lens": np.array(wls, np.int64), "seq_lens": np.array([len(s) for s in seqs_c], np.int32)}
    smax = max(len(s) for s in seqs_c)
    tkp = np.full((Bc, smax), 66, np.int32)
    gkp = np.full((Bc, smax), 16, np.int32)
    for b, sq in enumerate(seqs_c):
        tkp[b, :len(sq)] = sq
        g = alc._map_phonemes_to_groups(sq)
        gkp[b, :len(sq)] = g.tolist() if isinstance(g, torch.Tensor) else list(g)
    z["tokens"], z["group_tokens"] = tkp, gkp
    n_est = 0
    for b in range(Bc):
        for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
            rows = resc[b][key]
            z[f"{short}{b}_int"] = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
            z[f"{short}{b}_flt"] = np.array([[float(r[5]), float(r[6]), float(r[7])] for r in rows], np.float32).reshape(-1, 3)
            n_est += int(z[f"{short}{b}_int"][:, 4].sum())
    np.savez_compressed(os.path.join(HERE, "l2_complete.npz"), **z)
    print("l2_complete: estimated rows", n_est, "of", sum(len(s) for s in seqs_c) * 2)

    path = os.path.join(HERE, "host_cases.json")
    json.dump(out, open(path, "w"), ensure_ascii=False)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
