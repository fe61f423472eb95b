"""Live fuzz of the oracle against the reference itself (forced_alignment.py / utils.py imported by
file path).  Runs only where /root/reference exists (the build container); the committed golden
vectors (tests/golden/) carry the same pin to machines without it."""
import numpy as np
import pytest
import torch

import cases
import refload

pytestmark = pytest.mark.skipif(not refload.available(), reason="/root/reference not present")


@pytest.mark.parametrize("seed", [0, 1])
def test_decode_alignments_fuzz(ora, seed):
    torch.set_num_threads(1)
    fa = refload.forced_alignment()
    rng = np.random.default_rng(seed)
    n_seg = 0
    for it in range(30):
        C = 67 if rng.random() < 0.8 else 17
        blank = C - 1
        T = int(rng.integers(5, 260))
        kind = rng.integers(0, 4)
        S = int(rng.integers(1, max(2, T // 4))) if kind else int(rng.integers(max(1, T // 2), T + 2))
        peak = float(rng.choice([9.0, 4.0, 2.0, 0.5]))
        sil_rate = float(rng.choice([0.0, 0.15, 0.3]))
        anchors = int(rng.choice([10, 3, 0, 5]))
        tf, ign = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        boost, enf = bool(rng.random() < 0.8), bool(rng.random() < 0.8)
        lp, toks, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sil_rate=sil_rate, sil_len=(4, 30),
                                         repeat_rate=0.1)
        au = fa.AlignmentUtils(blank, 0, silence_anchors=anchors, ignore_noise=ign, truly_forced=tf)
        try:
            ref = au.decode_alignments(torch.from_numpy(lp)[None], torch.from_numpy(toks)[None], torch.tensor([T]),
                                       torch.tensor([S]), boost_targets=boost, enforce_minimum=enf)
        except ValueError:
            ref = None
        res = ora.decode_alignments(lp[None], toks[None], [T], [S], ora.make_params(blank, 0, anchors, ign, tf, boost, enf))
        if ref is None:
            assert res["status"][0] == ora.ERR_TOO_SHORT
        else:
            assert res["status"][0] == 0
            assert ora.segments_as_lists(res)[0] == [tuple(int(v) for v in x) for x in ref[0]]
            n_seg += int(res["mode"][0] == ora.MODE_SEGMENTED)
    assert n_seg >= 1


def test_simple_and_confidences_fuzz(ora):
    torch.set_num_threads(1)
    fa, ut = refload.forced_alignment(), refload.utils()
    rng = np.random.default_rng(9)
    for it in range(20):
        T = int(rng.integers(8, 300))
        S = int(rng.integers(1, max(2, int(T * 0.45))))
        lp, toks, _ = cases.planted_case(rng, T, S, C=67, peak=float(rng.choice([9.0, 2.0, 0.5])), repeat_rate=0.1)
        tf = bool(rng.integers(0, 2))
        au = fa.AlignmentUtils(66, 0, silence_anchors=0, truly_forced=tf)
        lpt = torch.from_numpy(lp)
        ref = au.decode_alignments_simple(lpt[None], torch.from_numpy(toks)[None], torch.tensor([T]), torch.tensor([S]))
        res = ora.decode_alignments(lp[None], toks[None], [T], [S], ora.make_params(66, 0, 0, True, tf, False, False), simple=True)
        assert ora.segments_as_lists(res)[0] == [tuple(int(v) for v in x) for x in ref[0]]
        fs = [(ph, max(0, s - 1), min(T + 1, e + 2), idx, False) for (ph, s, e, idx) in ref[0]]
        if fs:
            want = np.array([x[5] for x in ut._calculate_confidences(lpt, fs)], np.float32)
            rc, conf, _, _ = ora.confidences(lp, fs)
            assert rc == 0
            np.testing.assert_allclose(conf, want, atol=2e-7, rtol=0)


def test_level2_fuzz(ora):
    """>= 500 utterances through the REFERENCE's extract_timestamps_from_segment_batch (stub model, both heads,
    core.py:811-992) against the oracle's chain ensure_target_coverage_default -> extend_soft_boundaries ->
    confidences -> convert_to_ms: start / end frames, target indices and ms bit-exact, confidences <= 1.2e-7."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_l2 import make_batch
    from test_oracle_golden import oracle_level2
    torch.set_num_threads(1)
    rng = np.random.default_rng(4242)
    n_utt = n_rows = n_exact = n_ext = 0
    for it in range(32):
        soft = int(rng.choice([3, 3, 3, 7, 1, 5]))
        al = refload.core_aligner(boundary_softness=soft)
        al.warn_level = 0
        B, Tpad = 16, int(rng.choice([160, 260]))
        peaks = [[8.0, 6.0], [7.0, 4.0], [5.0, 2.5], [3.0, 2.0]][it % 4]
        lc, lg, spec, wav_lens, seqs = make_batch(al, rng, B, Tpad, peaks, float(rng.choice([0.0, 0.15, 0.25])),
                                                  flat_every=int(rng.choice([0, 3, 5])))
        al._cupe_prediction_batch = lambda wavs, wl, ee, _r=(lc, lg, spec): (_r[0], _r[1], None, list(_r[2]))
        al.extractor = object()
        offs = [round(0.5 * b, 2) for b in range(B)]
        res, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), wav_lens, [list(s) for s in seqs],
                                                             start_offset_times=offs, extract_embeddings=False,
                                                             do_groups=True, debug=False)
        smax = max(len(s) for s in seqs)
        tk = np.full((B, smax), 66, np.int32)
        gk = np.full((B, smax), 16, np.int32)
        for b, s in enumerate(seqs):
            tk[b, :len(s)] = s
            g = al._map_phonemes_to_groups(s)
            gk[b, :len(s)] = g.tolist() if isinstance(g, torch.Tensor) else list(g)
        slens = np.array([len(s) for s in seqs], np.int32)
        for key, logits, toks, blank in (("phoneme_timestamps", lc.numpy(), tk, 66), ("group_timestamps", lg.numpy(), gk, 16)):
            got = oracle_level2(ora, logits, toks, slens, spec, wav_lens, offs, blank, soft)
            for b, (rows, conf, sm, em) in enumerate(got):
                ref = res[b][key]
                ri = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in ref], np.int32).reshape(-1, 5)
                rf = np.array([[float(r[5]), float(r[6]), float(r[7])] for r in ref], np.float32).reshape(-1, 3)
                np.testing.assert_array_equal(rows, ri, err_msg=f"iteration {it} {key} item {b}")
                np.testing.assert_allclose(conf, rf[:, 0], atol=1.2e-7, rtol=0)
                np.testing.assert_array_equal(np.stack([sm, em], 1), rf[:, 1:])
                n_rows += len(conf)
                n_exact += int((conf.view(np.int32) == rf[:, 0].copy().view(np.int32)).sum())
        n_utt += B
    assert n_utt >= 500 and n_rows > 10000
    assert n_exact > 0.9 * n_rows, (n_exact, n_rows)


def test_threshold_adjacent_exponentials(ora):
    """Decisions that ride on torch.exp: `>= 0.9` / `>= 0.8` (silence windows, forced_alignment.py:514), `> 0.1` and
    `> half` (confidences, utils.py:99-103), `>= 1e-3` / `>= 10^-softness` (soft boundaries, core.py:724-803).
    torch.exp on a CPU tensor is MKL VML (not restatable); the oracle and the kernels use the correctly rounded
    exponential instead.  For every threshold, EVERY float32 argument whose exponential lies within +-4 ulp of the
    threshold is enumerated and the decision compared with torch's: a flip needs the two exponentials to differ AND to
    straddle the threshold.  The measured flip counts are the honest bound on 'bit-exact boundaries' for adversarial
    (threshold-adjacent) posteriors; they are recorded in DESIGN.md."""
    import math
    total = flips = differing = sleef_differing = sleef_flips = 0
    for thr in (0.9, 0.8, 0.1, 1e-3, 1e-7, 1e-1, 1e-5, 0.5, 0.25):
        t32 = np.float32(thr)
        x0 = np.float32(math.log(float(t32)))
        # arguments around log(thr): walk the float32 grid both ways until the exponential is > 4 ulp away
        xs = [x0]
        for direction in (np.float32(-np.inf), np.float32(np.inf)):
            x = x0
            for _ in range(400):
                x = np.nextafter(x, direction)
                xs.append(x)
        xs = np.array(sorted(set(float(v) for v in xs)), np.float32)
        mine = ora.exp_cr(xs)
        tor = torch.exp(torch.from_numpy(xs)).numpy()
        ulp = np.spacing(t32)
        near = np.abs(mine.astype(np.float64) - float(t32)) <= 4 * float(ulp)
        assert near.sum() >= 1  # (small thresholds: one float32 step of the argument moves the exponential by many ulp)
        d_ge = (mine[near] >= t32) != (tor[near] >= t32)
        d_gt = (mine[near] > t32) != (tor[near] > t32)
        total += int(near.sum())
        differing += int((mine[near].view(np.int32) != tor[near].view(np.int32)).sum())
        flips += int(d_ge.sum()) + int(d_gt.sum())
        sl = ora.expf_u10(xs)  # what round 1 used in these passes
        sleef_differing += int((sl[near].view(np.int32) != tor[near].view(np.int32)).sum())
        sleef_flips += int(((sl[near] >= t32) != (tor[near] >= t32)).sum()) + int(((sl[near] > t32) != (tor[near] > t32)).sum())
        assert np.abs(mine[near].view(np.int32).astype(np.int64) - tor[near].view(np.int32)).max() <= 1
    # silence detection end to end on crafted rows: k identical frames whose P(SIL) sits on the 0.9 line
    fa = refload.forced_alignment()
    vd = fa.ViterbiDecoder(blank_id=66, silence_id=0, silence_anchors=3)
    det_total = det_flip = 0
    x0 = np.float32(math.log(float(np.float32(0.9))))
    x = x0
    for step in range(-40, 41):
        xv = x0
        for _ in range(abs(step)):
            xv = np.nextafter(xv, np.float32(np.inf if step > 0 else -np.inf))
        lp = np.full((12, 67), -8.0, np.float32)
        lp[3:9, 0] = xv
        ref = vd._detect_silence_segments(torch.from_numpy(lp), sil_prob_threshold=0.9, min_silence_frames=3)
        got = ora.detect_silence(lp, 0, 0.9, 3)
        det_total += 1
        det_flip += int([tuple(int(v) for v in r) for r in ref] != got)
    print(f"threshold-adjacent: {total} arguments within 4 ulp of a threshold, {differing} exponentials differ from "
          f"torch by 1 ulp, {flips} decisions flip (Sleef expf_u10 instead: {sleef_differing} differ, {sleef_flips} flip); "
          f"silence windows on the 0.9 line: {det_flip}/{det_total} differ")
    # the correctly rounded restatement keeps the adversarial flip rate low; it cannot be zero without VML itself
    assert flips <= 0.05 * 2 * total
    assert det_flip <= 0.1 * det_total


def test_strided_mean_is_torchs(ora):
    """core.py:711 `probs[start:end, phoneme].mean()`: torch's float32 mean of a strided view is ATen's cascade sum (four
    interleaved accumulators, a level per sixteen rows), divided by n in float32.  The oracle's restatement gives the same
    bits as torch on every random column (lengths 1..700: all cascade levels); a running float32 sum and a float64
    accumulation -- what rounds 1-4 used -- do not."""
    torch.set_num_threads(1)
    rng = np.random.default_rng(31)
    n_cases = n_f64_differs = n_seq_differs = 0
    for _ in range(1500):
        n = int(rng.integers(1, 700))
        lp = np.log(np.clip(rng.random((n + 6, 67)).astype(np.float32) ** 3, 1e-30, 1)).astype(np.float32)
        ph, s = int(rng.integers(0, 67)), 3
        probs = torch.from_numpy(ora.exp_cr(lp))          # the oracle's exponentials; torch's own summation on top of them
        want = np.float32(probs[s:s + n, ph].mean().item())
        got = ora.strided_mean(lp, s, s + n, ph)
        assert got.view(np.int32) == want.view(np.int32), (n, got, want)
        col = probs[s:s + n, ph].numpy()
        n_f64_differs += int(np.float32(col.astype(np.float64).sum() / n) != want)
        acc = np.float32(0)
        for v in col:
            acc = np.float32(acc + v)
        n_seq_differs += int(np.float32(acc / np.float32(n)) != want)
        n_cases += 1
    print(f"strided mean: {n_cases} columns equal to torch; a float64 accumulation would differ on {n_f64_differs}, a running "
          f"float32 sum on {n_seq_differs}")
    assert n_f64_differs > 0 and n_seq_differs > 0


def test_sliding_mean_threshold_adjacency(ora):
    """_detect_silence_segments (forced_alignment.py:503-517): avg = float32(cumsum[i+k-1] - cumsum[i-1]) / k >= 0.9 on
    P(SIL) = torch.exp(...).  torch.exp differs from the correctly rounded exponential by 1 ulp on ~1 % of arguments; this
    enumerates windows whose mean sits on the 0.9 line AND that contain such an argument, runs the reference's detector and
    the oracle's on the same rows, and counts how often the 1-ulp difference flips a silence run.  The count is the honest
    bound on 'bit-exact boundaries in the silence-anchored mode' for threshold-adjacent posteriors."""
    import math
    torch.set_num_threads(1)
    fa = refload.forced_alignment()
    # arguments with P(SIL) in [0.6, 1): the frames a window on the 0.9 line is made of
    rng = np.random.default_rng(77)
    xs = np.unique((-rng.random(400000) * 0.51).astype(np.float32))
    mine, tor = ora.exp_cr(xs), torch.exp(torch.from_numpy(xs)).numpy()
    assert np.abs(mine.view(np.int32).astype(np.int64) - tor.view(np.int32)).max() <= 1
    differing = xs[mine.view(np.int32) != tor.view(np.int32)]
    frac = len(differing) / len(xs)
    assert 0 < frac < 0.05
    windows = near = flips = 0
    thr = np.float32(0.9)
    for k in (3, 10):
        vd = fa.ViterbiDecoder(blank_id=66, silence_id=0, silence_anchors=k)
        for xd in differing[:400]:
            pd_t = float(torch.exp(torch.tensor([xd]))[0])
            py0 = (float(thr) * k - pd_t) / (k - 1)   # the other k - 1 frames put the window's mean on the line
            if not (0.0 < py0 < 1.0):
                continue
            y0 = np.float32(math.log(py0))
            ys = [y0]
            for d in (np.float32(-np.inf), np.float32(np.inf)):
                y = y0
                for _ in range(6):
                    y = np.nextafter(y, d)
                    ys.append(y)
            for y in ys:
                if y > 0:
                    continue
                # a run of k frames: one frame with the differing exponential, k - 1 frames at y; no silence anywhere else
                lp = np.full((k + 8, 67), -20.0, np.float32)
                lp[4:4 + k, 0] = y
                lp[4 + k // 2, 0] = xd
                windows += 1
                py = float(torch.exp(torch.tensor([y]))[0])
                mean = (pd_t + (k - 1) * py) / k
                if abs(mean - float(thr)) > 4 * float(np.spacing(thr)):
                    continue
                near += 1
                ref = vd._detect_silence_segments(torch.from_numpy(lp), sil_prob_threshold=0.9, min_silence_frames=k)
                got = ora.detect_silence(lp, 0, 0.9, k)
                flips += int([tuple(int(v) for v in r) for r in ref] != got)
    print(f"sliding-mean threshold: {len(differing)} of {len(xs)} arguments with P(SIL) in [0.6, 1) ({100 * frac:.2f} %) have "
          f"exponentials that differ from torch's by 1 ulp; of {windows} windows built around the first 400 of them {near} "
          f"have a mean within 4 ulp of 0.9, {flips} of those flip a silence run")
    assert near > 100
    assert flips <= 0.5 * near
