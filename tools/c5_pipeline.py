#!/usr/bin/env python3
"""tools/c5_pipeline.py -- BASELINE.json configs[4] harness: a directory of posterior dumps -> device pipeline -> TextGrids.

The real run (LJSpeech-13k through the cupe2i model on PyTorch-ROCm) is blocked on the checkpoint, which is not
available offline -- not on code: this script is everything behind the model.  Input directory, one file per utterance:

    <utt>.npz   logits_class [T,67] f32   raw outputs of the phoneme head (core.py:897; log-softmax happens on the device)
                logits_group [T,17] f32   raw outputs of the group head
                ph66 [S] int              target ids; SIL (0) where the text has punctuation (ph66_phonemeizer.py:185-199),
                                          which is what sends real text through the silence-anchored mode
                pg16 [S] int              group ids of the same targets
                wav_len int               samples of the 16 kHz clip (for the frame -> ms conversion, utils.py:115-149)
                words / word_num / eipa   optional: word list, word index per target, ipa label per target

    python tools/c5_pipeline.py run IN_DIR OUT_DIR [--batch 256] [--device cuda:0] [--confidence]
        utterances sorted by length, batched, both heads through PhonemeTimestampAligner.extract_timestamps_from_logits
        (align -> coverage -> soft boundaries -> confidences -> ms, all on the device), then post_process_segment
        (core.py:1140) and the TextGrid writer (utils.py:152-411): OUT_DIR/<utt>.TextGrid and OUT_DIR/<utt>.vs.json
    python tools/c5_pipeline.py diff DIR_A DIR_B
        byte-compare the TextGrids of two runs (e.g. this pipeline against the CPU reference's, or against the
        oracle-generated ones of tests/test_gpu_c5.py); exit status 1 on any difference
    python tools/c5_pipeline.py synth OUT_DIR [--n 64] [--seed 5]
        a synthetic input directory (planted paths with silences at ~1/10 of the targets; inputs only)
"""
import argparse
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_dir(path):
    utts = []
    for f in sorted(glob.glob(os.path.join(path, "*.npz"))):
        z = np.load(f, allow_pickle=False)
        u = {"name": os.path.splitext(os.path.basename(f))[0],
             "logits_class": z["logits_class"].astype(np.float32), "logits_group": z["logits_group"].astype(np.float32),
             "ph66": [int(x) for x in z["ph66"]], "pg16": [int(x) for x in z["pg16"]], "wav_len": int(z["wav_len"])}
        for k in ("words", "word_num", "eipa"):
            if k in z.files:
                u[k] = [x if k == "word_num" else str(x) for x in z[k].tolist()]
        utts.append(u)
    return utts


def batches(utts, batch):
    order = sorted(range(len(utts)), key=lambda i: -utts[i]["logits_class"].shape[0])
    for i in range(0, len(order), batch):
        yield [utts[j] for j in order[i:i + batch]]


def pad_batch(us):
    import torch
    B = len(us)
    Tmax = max(u["logits_class"].shape[0] for u in us)
    lc = np.zeros((B, Tmax, 67), np.float32)
    lg = np.zeros((B, Tmax, 17), np.float32)
    lc[:, :, 66] = 6.0  # padded frames look like noise (they are beyond spectral_len and never aligned)
    lg[:, :, 16] = 6.0
    for b, u in enumerate(us):
        T = u["logits_class"].shape[0]
        lc[b, :T], lg[b, :T] = u["logits_class"], u["logits_group"]
    return torch.from_numpy(lc), torch.from_numpy(lg), [u["logits_class"].shape[0] for u in us]


def segment_dict(al, u, ts_rows):
    """post_process_segment (core.py:1140-1210) for one utterance -> the dict process_sentence returns"""
    dur = u["wav_len"] / al.resampler_sample_rate
    seg = {"start": 0.0, "end": dur, "text": u.get("text", u["name"]), "ph66": u["ph66"], "pg16": u["pg16"]}
    ts = {"ph66": u["ph66"], "pg16": u["pg16"], "eipa": u.get("eipa", [f"p{p}" for p in u["ph66"]]),
          "words": u.get("words", []), "word_num": u.get("word_num", [])}
    return {"segments": [al.post_process_segment(seg, ts, u["ph66"], ts_rows["phoneme_timestamps"],
                                                 ts_rows["group_timestamps"])]}


def run(in_dir, out_dir, batch=256, device="cuda:0", confidence=False, silence_anchors=10):
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    from bournemouth_forced_aligner_amd.textgrid import dict_to_textgrid
    os.makedirs(out_dir, exist_ok=True)
    utts = load_dir(in_dir)
    al = PhonemeTimestampAligner(preset=None, device=device, silence_anchors=silence_anchors,
                                 group_id_to_label={i: f"g{i}" for i in range(17)},
                                 phoneme_id_to_label={i: f"p{i}" for i in range(67)})
    n_seg = 0
    for us in batches(utts, batch):
        lc, lg, spec = pad_batch(us)
        rows = al.extract_timestamps_from_logits(lc, lg, spec, [u["ph66"] for u in us], [u["wav_len"] for u in us],
                                                 start_offset_times=0.0, group_sequences=[u["pg16"] for u in us])
        for u, r in zip(us, rows):
            d = segment_dict(al, u, r)
            with open(os.path.join(out_dir, u["name"] + ".TextGrid"), "w", encoding="utf-8") as f:
                f.write(dict_to_textgrid(d, include_confidence=confidence))
            with open(os.path.join(out_dir, u["name"] + ".vs.json"), "w", encoding="utf-8") as f:
                json.dump(d, f, ensure_ascii=False)
            n_seg += 1
    return n_seg


def diff_dirs(a, b):
    names = sorted(set(os.path.basename(f) for f in glob.glob(os.path.join(a, "*.TextGrid"))) |
                   set(os.path.basename(f) for f in glob.glob(os.path.join(b, "*.TextGrid"))))
    bad = []
    for n in names:
        fa, fb = os.path.join(a, n), os.path.join(b, n)
        if not (os.path.exists(fa) and os.path.exists(fb)):
            bad.append((n, "missing"))
        elif open(fa, encoding="utf-8").read() != open(fb, encoding="utf-8").read():
            la, lb = open(fa, encoding="utf-8").read().split("\n"), open(fb, encoding="utf-8").read().split("\n")
            k = next((i for i, (x, y) in enumerate(zip(la, lb)) if x != y), min(len(la), len(lb)))
            bad.append((n, f"line {k + 1}: {la[k] if k < len(la) else '<eof>'!r} != {lb[k] if k < len(lb) else '<eof>'!r}"))
    return len(names), bad


def synth(out_dir, n=64, seed=5):
    """Planted-path logits with SIL targets at ~1/10 of the positions and planted 12-40-frame silences (inputs only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    for k in range(n):
        T = int(rng.integers(120, 900))
        S = int(rng.integers(6, max(7, T // 9)))
        lp, tk, planted = cases.planted_case(rng, T, S, C=67, peak=float(rng.choice([9.0, 7.0, 5.0])), sil_rate=0.1,
                                             sil_len=(12, 40))
        logits_c = (lp * np.float32(rng.choice([1.0, 1.3]))).astype(np.float32)  # log-probs are valid logits
        pg = np.where(planted == 66, 16, np.where(planted == 0, 0, 1 + planted % 15))
        logits_g = rng.normal(0, 1, size=(T, 17)).astype(np.float32)
        logits_g[np.arange(T), pg] += 6.0
        groups = [0 if t == 0 else 1 + int(t) % 15 for t in tk]
        words, word_num, w = [], [], -1
        for j, t in enumerate(tk):
            if j == 0 or tk[j - 1] == 0 or rng.random() < 0.25:
                w += 1
                words.append(f"w{w}")
            word_num.append(w)
        np.savez_compressed(os.path.join(out_dir, f"utt{k:04d}.npz"), logits_class=logits_c, logits_group=logits_g,
                            ph66=np.asarray(tk, np.int32), pg16=np.asarray(groups, np.int32),
                            wav_len=np.int64(T * 320 + int(rng.integers(0, 320))), words=np.asarray(words),
                            word_num=np.asarray(word_num, np.int32), eipa=np.asarray([f"i{int(t)}" for t in tk]))
    return n


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("in_dir")
    r.add_argument("out_dir")
    r.add_argument("--batch", type=int, default=256)
    r.add_argument("--device", default="cuda:0")
    r.add_argument("--confidence", action="store_true")
    d = sub.add_parser("diff")
    d.add_argument("a")
    d.add_argument("b")
    s = sub.add_parser("synth")
    s.add_argument("out_dir")
    s.add_argument("--n", type=int, default=64)
    s.add_argument("--seed", type=int, default=5)
    args = ap.parse_args()
    if args.cmd == "run":
        print(f"{run(args.in_dir, args.out_dir, args.batch, args.device, args.confidence)} TextGrids written to {args.out_dir}")
    elif args.cmd == "synth":
        print(f"{synth(args.out_dir, args.n, args.seed)} synthetic utterances written to {args.out_dir}")
    else:
        n, bad = diff_dirs(args.a, args.b)
        for name, why in bad:
            print(f"DIFF {name}: {why}")
        print(f"{n - len(bad)} / {n} TextGrids identical")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
