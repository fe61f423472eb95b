#!/usr/bin/env python3
"""tests/soak.py [n_batches] [seed] [--record file.json] -- randomized differential soak of the HIP path against the CPU
oracle (run on the GPU box).  Every batch draws its own head width, flags, floor probability, length ranges and posterior
sharpness; integer outputs must be identical.  `run()` is what `pytest -m gpu` calls on a fixed-seed slice
(tests/test_gpu_parity.py::test_soak_slice); the full soak takes minutes, prints one line per mismatch and a summary, and
with --record appends its per-seed counts to a JSON file (profiles/r04_soak.json).  A third of the batches hold 64-150
utterances: mixed-length calls through the one-kernel path (k_mix)."""
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


def run(nb=200, seed=1, dev=None):
    """Returns {"seed", "batches", "utterances", "mismatching", "post_dp_mismatches", "fused_mismatches", "seconds"}."""
    rng = np.random.default_rng(seed)
    dev = dev or torch.device("cuda", 0)
    bad = 0
    bad2 = 0
    bad3 = 0
    items = 0
    t0 = time.time()
    for it in range(nb):
        C = int(rng.choice([67, 67, 67, 17, 17, 40, 12, 5]))  # (12, 5: rows shorter than one host vector -- torch's sequential softmax sum)
        blank = C - 1
        anchors = int(rng.choice([0, 0, 10, 3]))
        tf = bool(rng.integers(0, 2))
        ign = bool(rng.integers(0, 4) != 0)
        simple = bool(rng.integers(0, 8) == 0)
        boost = bool(rng.integers(0, 6) != 0)
        enf = bool(rng.integers(0, 6) != 0)
        min_prob = float(rng.choice([1e-8, 1e-8, 1e-8, 1e-4, 0.05]))  # ViterbiDecoder.min_phoneme_prob (forced_alignment.py:20)
        regime = int(rng.integers(0, 7))
        one_lo, one_hi = [(15, 23), (33, 58), (58, 65)][int(rng.integers(0, 3))]  # regime 5: token counts of ONE window class (Rw = 1 / 2 / 3)
        n = int(rng.integers(4, 40))
        if regime == 6:  # paths beyond 1 024 states (the workgroup-wide kernel), now and then beyond 8 192 (its sixteen-wave form)
            n = int(rng.integers(1, 4))
        elif rng.integers(0, 3) == 0 and regime != 5:  # 64 utterances or more with different lengths: the one-kernel mixed-length path (k_mix)
            n = int(rng.integers(64, 150))
        lps, toks = [], []
        for _ in range(n):
            if regime == 0:      # headline-like: window classes
                S = int(rng.integers(15, 125)); T = int(rng.integers(4 * S + 1, min(1600, 12 * S) + 1))
            elif regime == 1:    # tight: T close to L, strides 1..4
                S = int(rng.integers(1, 200)); T = int(rng.integers(max(1, S - 2), 5 * S + 8))
            elif regime == 2:    # long
                S = int(rng.integers(60, 420)); T = int(rng.integers(4 * S + 1, 4 * S + 900))
            elif regime == 3:    # tiny
                S = int(rng.integers(0, 12)); T = int(rng.integers(1, 90))
            elif regime == 6:
                S = int(rng.integers(2080, 2400)) if rng.integers(0, 4) == 0 else int(rng.integers(260, 700))
                T = int(rng.integers(4 * S + 1, 4 * S + 400))
            elif regime == 5:    # a small call of one sliding-window class: plan + DP + rerun + walk in one kernel (k_one)
                S = int(rng.integers(one_lo, one_hi)); T = int(rng.integers(4 * S + 1, 4 * S + 700))
            else:                # mixed
                T = int(rng.integers(8, 1200)); S = int(rng.integers(1, max(2, T // 3)))
            peak = float(rng.choice([9.0, 6.0, 3.0, 1.0, 0.2]))
            sil_rate = float(rng.choice([0.0, 0.0, 0.08, 0.2])) if anchors > 0 else 0.0
            lp, tk, _ = cases.planted_case(rng, max(T, 1), S, C=C, blank=blank, peak=peak, sigma=float(rng.choice([1.0, 2.0])),
                                           sil_rate=sil_rate, repeat_rate=float(rng.choice([0.0, 0.15])))
            lps.append(lp); toks.append(tk)
        lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
        au = AlignmentUtils(blank, 0, silence_anchors=anchors, ignore_noise=ign, truly_forced=tf)
        au.viterbi_decoder.min_phoneme_prob = min_prob
        min_log = float(torch.log(torch.tensor(min_prob, dtype=torch.float32)))  # the float32 floor, forced_alignment.py:70
        au.viterbi_decoder.window_max_tokens = None if rng.integers(0, 2) else 4096
        hint = None if rng.integers(0, 2) else 0   # None: no hint at all; 0: derived by align_batch from the host lengths
        if rng.integers(0, 2):
            has_sil = bool((tk == 0).any())
            hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=has_sil, anchor_pauses=anchors > 0,
                                                      simple=simple, n_classes=C, boost_targets=boost, enforce_minimum=enf)
        # the boundary takes any row / batch stride (last dimension contiguous): pad rows and shift the base now and then
        lpd_in = torch.from_numpy(lp).to(dev)
        padc = int(rng.choice([0, 0, 1, 3, 5]))
        if padc or rng.integers(0, 3) == 0:
            shift = int(rng.integers(0, 4))
            big = torch.full((lp.shape[0], lp.shape[1] + 1, C + padc + shift), -7.0, dtype=torch.float32, device=dev)
            big[:, :lp.shape[1], shift:shift + C] = lpd_in
            lpd_in = big[:, :lp.shape[1], shift:shift + C]
            assert lpd_in.stride(2) == 1
        res = au.viterbi_decoder.align_batch(lpd_in, torch.from_numpy(tk), T_len, S_len,
                                             boost_targets=boost, enforce_minimum=enf, anchor_pauses=anchors > 0,
                                             simple=simple, seg_cap=lp.shape[1] + 1, class_mask=hint)
        torch.cuda.synchronize()
        prm = ora.make_params(blank, 0, anchors, ign, tf, boost, enf, min_log_prob=min_log)
        exp = ora.decode_alignments(lp, tk, T_len, S_len, prm, simple=simple)
        st = res.status.cpu().numpy()
        fph = res.frame_phonemes.cpu().numpy(); fidx = res.frame_phonemes_idx.cpu().numpy()
        cnt = res.seg_count.cpu().numpy(); segs = res.segs.cpu().numpy()
        for b in range(n):
            items += 1
            ok = st[b] == exp["status"][b]
            if ok and st[b] == 0:
                T = int(T_len[b])
                ok = (fph[b, :T] == exp["frame_ph"][b, :T]).all() and (fidx[b, :T] == exp["frame_idx"][b, :T]).all() \
                    and cnt[b] == exp["seg_count"][b] and (segs[b, :cnt[b]] == exp["seg"][b, :cnt[b]]).all()
            if not ok:
                bad += 1
                dump = os.path.join(ROOT, "gpurun_out", f"soak_s{seed}_b{it}_i{b}.npz")
                os.makedirs(os.path.dirname(dump), exist_ok=True)
                np.savez_compressed(dump, lp=lp[b, :int(T_len[b])], tk=tk[b, :int(S_len[b])], C=C, anchors=anchors, tf=tf,
                                    ign=ign, simple=simple, boost=boost, enf=enf, got_ph=fph[b], got_idx=fidx[b],
                                    exp_ph=exp["frame_ph"][b], exp_idx=exp["frame_idx"][b], mode=exp["mode"][b])
                print(f"MISMATCH batch {it} item {b}: C={C} T={int(T_len[b])} S={int(S_len[b])} anchors={anchors} tf={tf} ign={ign} "
                      f"simple={simple} boost={boost} enf={enf} min_prob={min_prob} hint={hint} status={st[b]}/{exp['status'][b]}", flush=True)
        # ---- post-DP stages on the GPU's own tuples: confidences (utils.py:70-113), then ensure_target_coverage
        # (default) + extend_soft_boundaries (core.py:925-931)
        if (st == 0).all() and n > 0:
            from bournemouth_forced_aligner_amd import calculate_confidences_batch
            from bournemouth_forced_aligner_amd.utils import postprocess_batch
            lpd = lpd_in
            conf, _cst = calculate_confidences_batch(lpd, res.segs, res.seg_count)
            conf = conf.cpu().numpy()
            soft = int(rng.choice([3, 2, 5]))
            segs2 = res.segs.clone()
            cnt2 = res.seg_count.clone()
            postprocess_batch(lpd, torch.from_numpy(np.asarray(S_len, np.int32)), segs2, cnt2, extend=True, boundary_softness=soft)
            torch.cuda.synchronize()
            segs2 = segs2.cpu().numpy(); cnt2 = cnt2.cpu().numpy()
            for b in range(n):
                tup = [tuple(int(v) for v in r) for r in segs[b, :cnt[b]]]
                rc, c, _s, _e = ora.confidences(lp[b], tup)
                okc = rc == 0 and (conf[b, :cnt[b]].view(np.int32) == c.view(np.int32)).all()
                cov = ora.ensure_target_coverage_default(tup, int(S_len[b]))
                ext = ora.extend_soft_boundaries(lp[b], cov, soft) if cov else []
                got = [tuple(int(v) for v in r) for r in segs2[b, :cnt2[b]]]
                okp = got == ext
                if not (okc and okp):
                    bad2 += 1
                    print(f"POST-DP MISMATCH batch {it} item {b}: C={C} T={int(T_len[b])} S={int(S_len[b])} conf_ok={okc} post_ok={okp} "
                          f"softness={soft}", flush=True)
        # ---- fused front end (bfa_align_heads): the same batch as RAW logits (scaled / shifted log-probs) against the
        # two-pass path (bfa_log_softmax, then bfa_align_batch) -- states, tuples, status, then confidences and the
        # soft-boundary stage from (logits, row statistics); every array bitwise
        if not simple and n > 0 and it % 2 == 0 and lp.shape[1] + 1 <= 6000:
            from bournemouth_forced_aligner_amd import calculate_confidences_batch, log_softmax
            from bournemouth_forced_aligner_amd.forced_alignment import align_heads
            from bournemouth_forced_aligner_amd.utils import postprocess_batch
            logits = torch.from_numpy(lp).to(dev) * float(rng.choice([1.0, 1.7, 0.6])) + float(rng.normal(0, 2))
            lp2 = log_softmax(logits)
            two = au.viterbi_decoder.align_batch(lp2, torch.from_numpy(tk), T_len, S_len, boost_targets=boost,
                                                 enforce_minimum=enf, anchor_pauses=anchors > 0, seg_cap=lp.shape[1] + 1)
            (fus, stats), = align_heads([au], [logits], [torch.from_numpy(tk)], T_len, S_len, boost_targets=boost,
                                        enforce_minimum=enf, seg_cap=lp.shape[1] + 1)
            torch.cuda.synchronize()
            same = torch.equal(two.status, fus.status) and torch.equal(two.seg_count, fus.seg_count)
            if same and bool((two.status == 0).all()):
                mk = torch.arange(two.segs.shape[1], device=dev)[None, :] < two.seg_count[:, None]
                same = torch.equal(two.segs[mk], fus.segs[mk]) and \
                    torch.equal(two.frame_phonemes, fus.frame_phonemes) and torch.equal(two.frame_phonemes_idx, fus.frame_phonemes_idx)
                if same:
                    c1, _ = calculate_confidences_batch(lp2, two.segs, two.seg_count)
                    c2, _ = calculate_confidences_batch(logits, fus.segs, fus.seg_count, row_stats=stats)
                    s1, n1 = two.segs.clone(), two.seg_count.clone()
                    s2, n2 = fus.segs.clone(), fus.seg_count.clone()
                    Sd = torch.from_numpy(np.asarray(S_len, np.int32))
                    postprocess_batch(lp2, Sd, s1, n1, extend=True, boundary_softness=3)
                    postprocess_batch(logits, Sd, s2, n2, extend=True, boundary_softness=3, row_stats=stats)
                    torch.cuda.synchronize()
                    m = torch.arange(c1.shape[1], device=dev)[None, :] < two.seg_count[:, None]
                    same = torch.equal(c1.view(torch.int32)[m], c2.view(torch.int32)[m]) and torch.equal(n1, n2)
                    if same:
                        m2 = torch.arange(s1.shape[1], device=dev)[None, :] < n1[:, None]
                        same = torch.equal(s1[m2], s2[m2])
            if not same:
                bad3 += 1
                why = []
                if not torch.equal(two.status, fus.status): why.append(f"status {two.status.tolist()} vs {fus.status.tolist()}")
                elif not torch.equal(two.seg_count, fus.seg_count): why.append("seg_count")
                elif not torch.equal(two.frame_phonemes, fus.frame_phonemes): 
                    d = (two.frame_phonemes != fus.frame_phonemes).nonzero()
                    why.append(f"frames differ at {d[:4].tolist()} T_len={[int(T_len[int(i)]) for i in d[:4, 0]]} Tmax={lp.shape[1]}")
                else: why.append("segs/conf/post")
                print(f"FUSED-vs-TWO-PASS MISMATCH batch {it}: C={C} anchors={anchors} tf={tf} ign={ign} boost={boost} enf={enf} regime={regime} {why}", flush=True)
    print(f"soak: {items} utterances in {nb} batches, {bad} mismatching, {bad2} post-DP mismatches, "
          f"{bad3} fused-front-end mismatches, {time.time() - t0:.0f} s (seed {seed})")
    return {"seed": seed, "batches": nb, "utterances": items, "mismatching": bad, "post_dp_mismatches": bad2,
            "fused_mismatches": bad3, "seconds": round(time.time() - t0, 1)}


def main():
    argv = [a for a in sys.argv[1:]]
    record = None
    if "--record" in argv:
        k = argv.index("--record")
        record = argv[k + 1]
        del argv[k:k + 2]
    nb = int(argv[0]) if len(argv) > 0 else 200
    seed = int(argv[1]) if len(argv) > 1 else 1
    out = run(nb, seed)
    if record:
        import json
        try:
            with open(record) as f:
                doc = json.load(f)
        except Exception:
            doc = {"what": "tests/soak.py: randomized differential soak of the HIP path against the CPU oracle, per seed",
                   "runs": []}
        doc["runs"].append(out)
        doc["total_utterances"] = sum(r["utterances"] for r in doc["runs"])
        doc["total_mismatches"] = sum(r["mismatching"] + r["post_dp_mismatches"] + r["fused_mismatches"] for r in doc["runs"])
        os.makedirs(os.path.dirname(os.path.abspath(record)), exist_ok=True)
        with open(record, "w") as f:
            json.dump(doc, f, indent=1)
    return 1 if (out["mismatching"] or out["post_dp_mismatches"] or out["fused_mismatches"]) else 0


if __name__ == "__main__":
    sys.exit(main())
