"""-m gpu : parity of the HIP path (through the C-ABI) against the CPU oracle on identical inputs.
Integer outputs (framewise states, boundaries, target indices) must be bit-exact; confidences are
compared bit-exact as well (both sides use the same restated float32 exp), with the north-star
tolerance 1e-4 as the hard bound."""
import os
import sys

import numpy as np
import pytest
import torch

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _mk_batch(rng, n, C, Tr, Sr, **kw):
    blank = C - 1
    lps, toks = [], []
    for _ in range(n):
        T = int(rng.integers(*Tr))
        S = int(rng.integers(Sr[0], max(Sr[0] + 1, min(Sr[1], T))))
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, **kw)
        lps.append(lp)
        toks.append(tk)
    return cases.pad_batch(lps, toks, C, blank)


def _run_both(ora, dev, lp, tk, T_len, S_len, C, anchors=10, ign=True, tf=True, boost=True, enf=True, simple=False,
              window_max_tokens=None):
    from bournemouth_forced_aligner_amd import AlignmentUtils
    blank = C - 1
    au = AlignmentUtils(blank, 0, silence_anchors=anchors, ignore_noise=ign, truly_forced=tf)
    au.viterbi_decoder.window_max_tokens = window_max_tokens
    lpd = torch.from_numpy(lp).to(dev)
    res = au.viterbi_decoder.align_batch(lpd, torch.from_numpy(tk), T_len, S_len, boost_targets=boost,
                                         enforce_minimum=enf, anchor_pauses=anchors > 0, simple=simple,
                                         seg_cap=lp.shape[1] + 1)
    torch.cuda.synchronize()
    prm = ora.make_params(blank, 0, anchors, ign, tf, boost, enf)
    exp = ora.decode_alignments(lp, tk, T_len, S_len, prm, simple=simple)
    return res, exp


def _compare(res, exp, T_len, check_mode=True):
    st = res.status.cpu().numpy()
    np.testing.assert_array_equal(st, exp["status"])
    cnt = res.seg_count.cpu().numpy()
    segs = res.segs.cpu().numpy()
    fph = res.frame_phonemes.cpu().numpy()
    fidx = res.frame_phonemes_idx.cpu().numpy()
    for b in range(len(st)):
        if st[b] != 0:
            continue
        T = int(T_len[b])
        np.testing.assert_array_equal(fph[b, :T], exp["frame_ph"][b, :T], err_msg=f"frame phonemes item {b}")
        np.testing.assert_array_equal(fidx[b, :T], exp["frame_idx"][b, :T], err_msg=f"frame idx item {b}")
        assert cnt[b] == exp["seg_count"][b], f"segment count item {b}"
        np.testing.assert_array_equal(segs[b, :cnt[b]], exp["seg"][b, :cnt[b]], err_msg=f"segments item {b}")
    if check_mode:
        md = res.mode.cpu().numpy()
        ok = st == 0
        np.testing.assert_array_equal(md[ok], exp["mode"][ok])


def test_log_softmax_bit_exact(ora, gpu_device):
    from bournemouth_forced_aligner_amd import log_softmax
    rng = np.random.default_rng(11)
    for C in (67, 17, 16, 33, 128):
        x = (rng.normal(0, 3, size=(1531, C))).astype(np.float32)
        x[:, rng.integers(0, C)] += 9
        got = log_softmax(torch.from_numpy(x).to(gpu_device)).cpu().numpy()
        exp = ora.log_softmax_rows(x)
        assert (got.view(np.int32) == exp.view(np.int32)).all(), f"C={C}"


@pytest.mark.parametrize("C", [67, 17])
@pytest.mark.parametrize("peak", [9.0, 2.0, 0.5])
def test_standard_mode_parity(ora, gpu_device, C, peak):
    rng = np.random.default_rng(100 + C + int(peak * 10))
    lp, tk, T_len, S_len = _mk_batch(rng, 48, C, (8, 420), (1, 90), peak=peak, sigma=1.0, repeat_rate=0.1)
    for tf in (True, False):
        res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=0, tf=tf)
        _compare(res, exp, T_len)


@pytest.mark.parametrize("C", [67, 17])
def test_sliding_window_classes(ora, gpu_device, C):
    """Standard-mode DPs whose band is narrower than the path run in the sliding-window consumer (Rw = 1..4,
    bfa_dp3.inc DpCoreW); sharp posteriors stay above the sentinel, flat ones end at it and are redone with the
    full layout.  Both must reproduce the oracle, for T from just above L (pace ~ 1) to T >> L."""
    rng = np.random.default_rng(4200 + C)
    blank = C - 1
    lps, toks = [], []
    for S in (15, 16, 20, 25, 26, 33, 40, 47, 48, 60, 63, 64, 90, 120, 121, 150, 180, 187, 188, 220, 250):   # L = 4S+1: window classes 1,2,3,4,6,8 and none
        for T in (4 * S + 1, 4 * S + 2, 4 * S + 9, 5 * S + 3, min(8 * S, 1536), 25 * S):
            for peak in (9.0, 3.0, 0.3):
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=1.0, repeat_rate=0.15)
                lps.append(lp)
                toks.append(tk)
    for lo in range(0, len(lps), 90):
        lp, tk, T_len, S_len = cases.pad_batch(lps[lo:lo + 90], toks[lo:lo + 90], C, blank)
        for tf in (True, False):
            for cap in (None, 4096):   # the library's token limit for the window, and no limit (all six classes)
                res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=0, tf=tf, window_max_tokens=cap)
                _compare(res, exp, T_len)


def test_split_consumer_classes(ora, gpu_device):
    """Full-layout classes R = 6 and 8 (65..127 tokens at stride 4); R = 8 runs with the DP split over two consumer
    waves (bfa_dp5.inc, per-frame lane masks, Item::split = 2; R = 6 too in BFA_SPLIT_R6 builds): sharp and flat
    posteriors (sentinel regime with wrapped backpointers), T from L to long utterances, T not a multiple of 16, both
    final-state rules."""
    rng = np.random.default_rng(515)
    C, blank = 67, 66
    lps, toks = [], []
    for S in (65, 80, 95, 96, 97, 110, 127):
        for T in (4 * S + 1, 4 * S + 17, 6 * S + 5, 1537, 2999):
            for peak in (9.0, 2.0, 0.3):
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=1.0, repeat_rate=0.15)
                lps.append(lp)
                toks.append(tk)
    for lo in range(0, len(lps), 35):
        lp, tk, T_len, S_len = cases.pad_batch(lps[lo:lo + 35], toks[lo:lo + 35], C, blank)
        for tf in (True, False):
            res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=0, tf=tf)
            _compare(res, exp, T_len)


def test_sliding_window_equals_full_layout(gpu_device):
    """The same batch through the window classes (hint bits 8-11) and through the full-layout classes only."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    rng = np.random.default_rng(77)
    C, blank = 67, 66
    lps, toks = [], []
    for _ in range(64):
        S = int(rng.integers(15, 121))
        T = int(rng.integers(4 * S + 1, 30 * S))
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=float(rng.choice([9.0, 2.0, 0.5])), sigma=1.0,
                                       repeat_rate=0.1)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    au = AlignmentUtils(blank, 0, silence_anchors=0)
    vd = au.viterbi_decoder
    vd.window_max_tokens = 4096
    h_win = vd.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=C)
    h_full = vd.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False)
    # (64 utterances of different lengths: the mixed-length path, whose narrow window classes are the EXACT ones, bits 20-23)
    assert ((h_win >> 8) | (h_win >> 20)) & 15 and not (((h_full >> 8) | (h_full >> 20)) & 15)
    lpd = torch.from_numpy(lp).to(gpu_device)
    out = []
    for h in (h_win, h_full, None):   # None: no hint at all, the library launches every class the shapes allow
        r = vd.align_batch(lpd, torch.from_numpy(tk), T_len, S_len, anchor_pauses=False, class_mask=h)
        torch.cuda.synchronize()
        out.append((r.frame_phonemes.cpu().numpy(), r.frame_phonemes_idx.cpu().numpy(), r.segs.cpu().numpy(),
                    r.seg_count.cpu().numpy()))
    for k in (1, 2):
        cnt = out[0][3]
        np.testing.assert_array_equal(cnt, out[k][3])
        for b in range(len(cnt)):
            T = int(T_len[b])
            np.testing.assert_array_equal(out[0][0][b, :T], out[k][0][b, :T])
            np.testing.assert_array_equal(out[0][1][b, :T], out[k][1][b, :T])
            np.testing.assert_array_equal(out[0][2][b, :cnt[b]], out[k][2][b, :cnt[b]])


def test_no_silence_hint_is_checked(gpu_device):
    """BFA_HINT_NO_SILENCE_TARGETS skips the silence planning; a target that contains the silence id then is an
    error status, not a silently different alignment."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    rng = np.random.default_rng(5)
    C, blank = 67, 66
    lp, tk, _ = cases.planted_case(rng, 300, 12, C=C, blank=blank, peak=9.0, sil_rate=0.0)
    lp2, tk2, _ = cases.planted_case(rng, 300, 12, C=C, blank=blank, peak=9.0, sil_rate=0.4)
    assert (tk2 == 0).any()
    lpb, tkb, T_len, S_len = cases.pad_batch([lp, lp2], [tk, tk2], C, blank)
    au = AlignmentUtils(blank, 0)
    hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=C)
    assert hint & _lib.HINT_NO_SILENCE_TARGETS
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lpb).to(gpu_device), torch.from_numpy(tkb), T_len, S_len,
                                         class_mask=hint)
    torch.cuda.synchronize()
    st = res.status.cpu().numpy()
    assert st[0] == _lib.ITEM_OK and st[1] == _lib.ITEM_BAD_HINT
    with pytest.raises(RuntimeError):
        res.raise_for_status()
    # a hint without the class an utterance needs is reported as well (here: only R = 16 for a path of 49 states)
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lpb[:1]).to(gpu_device), torch.from_numpy(tkb[:1]), T_len[:1],
                                         S_len[:1], class_mask=(1 << 6) | _lib.HINT_NO_SILENCE_TARGETS)
    torch.cuda.synchronize()
    assert res.status.cpu().numpy()[0] == _lib.ITEM_BAD_HINT


def test_ignore_noise_false_and_given_emissions(ora, gpu_device):
    rng = np.random.default_rng(5)
    lp, tk, T_len, S_len = _mk_batch(rng, 32, 67, (30, 300), (1, 40), peak=6.0)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0, ign=False, boost=False, enf=False)
    _compare(res, exp, T_len)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0, ign=False, boost=False, enf=True)
    _compare(res, exp, T_len)


def test_short_audio_modes(ora, gpu_device):
    """stride 3/2/1, proportional (T == S..) and the too-short error (T < S)."""
    rng = np.random.default_rng(6)
    lps, toks = [], []
    for (T, S) in [(20, 6), (20, 7), (20, 9), (20, 10), (20, 19), (20, 20), (12, 12), (9, 10), (5, 9), (1, 1), (2, 1)]:
        lp, tk, _ = cases.planted_case(rng, T, S, C=67, peak=5.0)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 67, 66)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0)
    _compare(res, exp, T_len)
    assert (res.status.cpu().numpy() == 1).sum() == 2


def test_simple_mode_parity(ora, gpu_device):
    rng = np.random.default_rng(7)
    lp, tk, T_len, S_len = _mk_batch(rng, 48, 67, (40, 400), (1, 60), peak=4.0)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0, simple=True, boost=False, enf=False)
    _compare(res, exp, T_len, check_mode=False)


def test_long_paths_R_classes(ora, gpu_device):
    """CTC paths of 65..1000 states exercise every states-per-lane class of K1."""
    rng = np.random.default_rng(8)
    lps, toks = [], []
    for S in (17, 33, 50, 64, 70, 100, 130, 200, 249):
        T = 4 * S + int(rng.integers(1, 60))
        lp, tk, _ = cases.planted_case(rng, T, S, C=67, peak=7.0)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 67, 66)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0)
    _compare(res, exp, T_len)


@pytest.mark.parametrize("C", [67, 17, 40])
def test_paths_beyond_1024_states(ora, gpu_device, C):
    """L > 1024 (more than 255 phonemes in one DP): the workgroup-wide K1 (k_dp_big) and its row-major backtrace.
    Mixed with short utterances in the same batch; default flags, truly_forced off, and the simple mode."""
    rng = np.random.default_rng(31 + C)
    blank = C - 1
    lps, toks = [], []
    for S, T in ((256, 1025), (300, 1300), (511, 2100), (600, 2000), (30, 200), (1023, 4093), (700, 2801), (255, 1021)):
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=float(rng.choice([8.0, 1.0])), repeat_rate=0.1)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    for kw in (dict(anchors=0), dict(anchors=0, tf=False), dict(anchors=0, simple=True), dict(anchors=10)):
        res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, **kw)
        _compare(res, exp, T_len, check_mode=not kw.get("simple", False))


def test_segmented_mode_with_pieces_beyond_1024_states(ora, gpu_device):
    """Silence-anchored mode whose speech segments are themselves longer than 1024 states."""
    rng = np.random.default_rng(77)
    C, blank = 67, 66
    lps, toks = [], []
    for S, T in ((620, 2700), (700, 3000), (400, 1700)):
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=9.0, sil_rate=0.004, sil_len=(14, 30))
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10)
    assert (exp["mode"] == 1).any(), "the case should reach the segmented mode"
    _compare(res, exp, T_len)


def test_soak_regressions(ora, gpu_device):
    """Cases the randomized soak (tests/soak.py) found: a silence segment whose SIL indices are spread over the whole
    segment although the concatenation is cut off at T; one with more SIL tokens than frames (empty index ranges); T < S
    where the segmented attempt succeeds, so that the "audio too short" error of the standard mode never fires."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_regressions.npz"))
    for name in ("sil_cut_at_T", "more_sil_tokens_than_frames", "segmented_before_too_short"):
        lp1, tk1 = z[name + "_lp"], z[name + "_tk"]
        C, anchors, tf, ign, simple, boost, enf = (int(v) for v in z[name + "_cfg"])
        lp, tk, T_len, S_len = cases.pad_batch([lp1], [tk1], C, C - 1)
        res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=anchors, ign=bool(ign), tf=bool(tf),
                             boost=bool(boost), enf=bool(enf), simple=bool(simple))
        assert exp["mode"][0] == 1
        _compare(res, exp, T_len)


def test_simple_mode_empty_target(ora, gpu_device):
    """decode_alignments_simple has no empty-target shortcut (forced_alignment.py:951-985): S = 0 runs the DP over the
    single blank state, and with ignore_noise=False a long blank run is reported."""
    rng = np.random.default_rng(3)
    C, blank = 67, 66
    lps, toks = [], []
    for T, S in ((17, 0), (64, 0), (5, 0), (40, 3)):
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=5.0)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    for ign in (False, True):
        res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=0, ign=ign, simple=True)
        _compare(res, exp, T_len, check_mode=False)
        if not ign:
            assert exp["seg_count"][0] == 1 and exp["seg_count"][1] == 1 and exp["seg_count"][2] == 0


def test_mixed_length_shard_parity(ora, gpu_device):
    """bench.py --ragged's generator (T ~ U[200,3000], S = T // 25, per-utterance planted paths, reference-default
    flags with the host class hint): every K1 class and layout in one batch, against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from bournemouth_forced_aligner_amd import AlignmentUtils
    C = 67
    lp, tk, T_len, S_len = bench.synth_ragged(96, 200, 3000, C, 1004, gpu_device)
    au = AlignmentUtils(blank_id=C - 1, silence_id=0)
    hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=C)
    res = au.decode_alignments_device(lp, tk, T_len.to(gpu_device), S_len.to(gpu_device), class_mask=hint)
    torch.cuda.synchronize()
    exp = ora.decode_alignments(lp.cpu().numpy(), tk.cpu().numpy(), T_len.numpy(), S_len.numpy(), ora.make_params(C - 1, 0),
                                seg_cap=res.segs.shape[1])
    _compare(res, exp, T_len.numpy())


def test_call_can_be_captured_into_a_graph(gpu_device):
    """bfa_align_batch only enqueues work (kernels, a memset, event fork/join onto the handle's streams): with device
    resident arguments the whole call can be stream-captured and replayed as a hipGraph."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    rng = np.random.default_rng(1)
    lps, toks = [], []
    for S, T in ((20, 300), (40, 700), (90, 1200), (10, 100), (120, 3000), (30, 500)):
        lp, tk, _ = cases.planted_case(rng, T, S, C=67, blank=66, peak=8.0, sil_rate=0.1 if S == 30 else 0.0)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 67, 66)
    vd = AlignmentUtils(66, 0).viterbi_decoder
    dev = gpu_device
    lpd = torch.from_numpy(lp).to(dev)
    tkd = torch.from_numpy(tk).to(dev).to(torch.int32)
    Td = torch.from_numpy(np.asarray(T_len, np.int32)).to(dev)
    Sd = torch.from_numpy(np.asarray(S_len, np.int32)).to(dev)
    ref = vd.align_batch(lpd, tkd, Td, Sd, class_mask=None)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # warm-up on a side stream, as torch asks for before a capture
        vd.align_batch(lpd, tkd, Td, Sd, class_mask=None)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = vd.align_batch(lpd, tkd, Td, Sd, class_mask=None)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert (res.status.cpu() == ref.status.cpu()).all()
    assert (res.frame_phonemes.cpu() == ref.frame_phonemes.cpu()).all()
    assert (res.frame_phonemes_idx.cpu() == ref.frame_phonemes_idx.cpu()).all()
    assert (res.seg_count.cpu() == ref.seg_count.cpu()).all()


def test_confidences_parity(ora, gpu_device):
    from bournemouth_forced_aligner_amd import calculate_confidences_batch
    rng = np.random.default_rng(9)
    lp, tk, T_len, S_len = _mk_batch(rng, 24, 67, (40, 300), (1, 40), peak=3.0)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=0)
    # widen the tuples so that ranges overlap neighbours (exercises the aliasing replay)
    segs = res.segs.cpu().numpy().copy()
    cnt = res.seg_count.cpu().numpy()
    for b in range(segs.shape[0]):
        for i in range(cnt[b]):
            segs[b, i, 1] = max(0, segs[b, i, 1] - int(rng.integers(0, 4)))
            segs[b, i, 2] = segs[b, i, 2] + int(rng.integers(0, 6))
    lpd = torch.from_numpy(lp).to(gpu_device)
    conf, status = calculate_confidences_batch(lpd, torch.from_numpy(segs), torch.from_numpy(cnt))
    conf = conf.cpu().numpy()
    for b in range(segs.shape[0]):
        rc, c, s, e = ora.confidences(lp[b], [tuple(r) for r in segs[b, :cnt[b]]])
        assert rc == 0
        np.testing.assert_allclose(conf[b, :cnt[b]], c, atol=1e-4, rtol=0)
        assert (conf[b, :cnt[b]].view(np.int32) == c.view(np.int32)).all()


def test_confidences_of_long_tuples(ora, gpu_device):
    """Tuples of a hundred to several hundred frames (what the soft-boundary passes leave on soft posteriors): the
    confidence pass takes them with the whole wave, 64 probes at a time, and adds in the reference's frame order
    (utils.py:95-108).  Disjoint tuples of distinct starts: no aliasing replay.  Bit-identical to the oracle."""
    from bournemouth_forced_aligner_amd import calculate_confidences_batch
    rng = np.random.default_rng(19)
    B, T, C = 12, 1400, 67
    lp = rng.standard_normal((B, T, C)).astype(np.float32)
    lp = lp - np.log(np.exp(lp.astype(np.float64)).sum(-1, keepdims=True)).astype(np.float32)
    cap = 24
    segs = np.zeros((B, cap, 4), np.int32)
    cnt = np.zeros(B, np.int32)
    for b in range(B):
        t, n = int(rng.integers(0, 5)), 0
        while n < cap:
            ln = int(rng.choice([3, 40, 97, 98, 130, 200, 333, 64 + 65]))
            if t + ln > T:
                break
            segs[b, n] = (int(rng.integers(0, C - 1)), t, t + ln, n)
            t += ln + int(rng.integers(0, 3))
            n += 1
        cnt[b] = n
    assert (segs[:, :, 2] - segs[:, :, 1]).max() > 300
    lpd = torch.from_numpy(lp).to(gpu_device)
    conf, status = calculate_confidences_batch(lpd, torch.from_numpy(segs), torch.from_numpy(cnt))
    conf = conf.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    for b in range(B):
        rc, c, s, e = ora.confidences(lp[b], [tuple(r) for r in segs[b, :cnt[b]]])
        assert rc == 0
        assert (conf[b, :cnt[b]].view(np.int32) == c.view(np.int32)).all()


@pytest.mark.parametrize("anchors", [10, 3, 5])
def test_segmented_mode_parity(ora, gpu_device, anchors):
    """Targets with SIL and planted silences: silence detection, matching, per-segment DPs with
    boundary padding and silence anchoring, silence fills -- and the fall-backs to standard mode."""
    rng = np.random.default_rng(300 + anchors)
    lps, toks = [], []
    for i in range(64):
        T = int(rng.integers(60, 500))
        S = int(rng.integers(3, max(4, T // 6)))
        lp, tk, _ = cases.planted_case(rng, T, S, C=67, peak=float(rng.choice([9.0, 5.0, 3.0])),
                                       sil_rate=float(rng.choice([0.1, 0.2, 0.35])), sil_len=(4, 40), repeat_rate=0.05)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 67, 66)
    for tf in (True, False):
        res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=anchors, tf=tf)
        _compare(res, exp, T_len)
    md = res.mode.cpu().numpy()
    assert (md == 1).sum() >= 5, "test inputs did not exercise the segmented mode"
    assert (md == 2).sum() >= 1


def test_segmented_mode_more_utterances_than_planners(ora, gpu_device):
    """k_plan_seg runs eight wavefronts per CU (2 048 workgroups), each taking several candidate utterances one after the
    other with the same LDS arrays (P(SIL), cumulative sums, run / group / segment / bucket scratch): a call with more
    silence-anchored candidates than planner workgroups, short utterances of mixed lengths so that the utterances a wave
    takes in a row differ in every array's fill."""
    rng = np.random.default_rng(977)
    lps, toks = [], []
    for i in range(2600):
        T = int(rng.integers(40, 140))
        S = int(rng.integers(3, max(4, T // 6)))
        lp, tk, _ = cases.planted_case(rng, T, S, C=67, peak=float(rng.choice([9.0, 5.0])),
                                       sil_rate=float(rng.choice([0.15, 0.3])), sil_len=(4, 24), repeat_rate=0.05)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 67, 66)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 67, anchors=5, tf=True)
    _compare(res, exp, T_len)
    md = res.mode.cpu().numpy()
    assert (md == 1).sum() >= 1500, "test inputs did not exercise the segmented mode"


def test_segmented_mode_group_head_and_flags(ora, gpu_device):
    rng = np.random.default_rng(41)
    lps, toks = [], []
    for i in range(40):
        T = int(rng.integers(80, 400))
        S = int(rng.integers(3, T // 6))
        lp, tk, _ = cases.planted_case(rng, T, S, C=17, peak=6.0, sil_rate=0.25, sil_len=(8, 30))
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, 17, 16)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 17, anchors=10)
    _compare(res, exp, T_len)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, 17, anchors=10, boost=False, enf=True, ign=False)
    _compare(res, exp, T_len)


@pytest.mark.parametrize("C", [67, 17])
def test_prepared_emissions_bit_exact(ora, gpu_device, C):
    """DP-level pin: the boosted / normalised / floored emissions (forced_alignment.py:121-129) the HIP path
    feeds its DP are bit-identical to the oracle's (which is bit-identical to torch's, see the golden tests)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    rng = np.random.default_rng(600 + C)
    lp, tk, T_len, S_len = _mk_batch(rng, 24, C, (5, 300), (1, 60), peak=float(rng.choice([9.0, 2.0, 0.5])), sigma=2.0)
    lp[3] *= 8.0  # widen the dynamic range: exp underflow / denormal results
    au = AlignmentUtils(C - 1, 0)
    for boost, enf in ((True, True), (True, False), (False, True)):
        got = au.viterbi_decoder.prepare_emissions(torch.from_numpy(lp).to(gpu_device), torch.from_numpy(tk), T_len,
                                                   S_len, boost_targets=boost, enforce_minimum=enf).cpu().numpy()
        prm = ora.make_params(C - 1, 0, 10, True, True, boost, enf)
        for b in range(lp.shape[0]):
            T, S = int(T_len[b]), int(S_len[b])
            rc, want = ora.prepare_emissions(lp[b, :T], tk[b, :S], prm)
            assert rc == 0
            assert (got[b, :T].view(np.int32) == want.view(np.int32)).all(), f"item {b} boost={boost} enf={enf}"


def test_bad_hint_on_a_wide_item(gpu_device):
    """A wide item (more than 256 states, full-layout class R >= 6) whose K1 class is missing from the caller's class
    mask is walked by nobody: it must come back as BFA_ITEM_BAD_HINT, not as OK with stale outputs (the per-class walk
    launches only looked for wide leftovers when window classes were in the mask)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    rng = np.random.default_rng(17)
    C, blank = 67, 66
    lp, tk, _ = cases.planted_case(rng, 1700, 130, C=C, blank=blank, peak=9.0)   # L = 521 -> class R = 12, no window (T > 1536)
    lp2, tk2, _ = cases.planted_case(rng, 300, 12, C=C, blank=blank, peak=9.0)   # L = 49  -> class R = 2
    lpb, tkb, T_len, S_len = cases.pad_batch([lp, lp2], [tk, tk2], C, blank)
    au = AlignmentUtils(blank, 0)
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lpb).to(gpu_device), torch.from_numpy(tkb), T_len, S_len,
                                         class_mask=(1 << 0) | _lib.HINT_NO_SILENCE_TARGETS)
    torch.cuda.synchronize()
    st = res.status.cpu().numpy()
    assert st[0] == _lib.ITEM_BAD_HINT and st[1] == _lib.ITEM_OK, st
    # and with the right hint both align
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lpb).to(gpu_device), torch.from_numpy(tkb), T_len, S_len)
    torch.cuda.synchronize()
    assert (res.status.cpu().numpy() == 0).all()


def test_soak_slice(gpu_device):
    """A fixed-seed slice of tests/soak.py (~2 000 utterances: every head width, flag, floor probability, length regime,
    class hint form and input stride the soak draws) in every `pytest -m gpu` run: alignment, post-DP stages and the
    fused front end against the oracle, 0 mismatches."""
    import soak
    out = soak.run(nb=90, seed=20260929, dev=gpu_device)
    assert out["utterances"] >= 1500, out
    assert out["mismatching"] == 0 and out["post_dp_mismatches"] == 0 and out["fused_mismatches"] == 0, out


def _stride4_path(tk, blank):
    S = len(tk)
    L = 4 * S + 1
    path = np.full(L, blank, np.int32)
    pidx = np.full(L, -1, np.int32)
    path[1::4] = tk
    pidx[1::4] = np.arange(S)
    return path, pidx, L


@pytest.mark.parametrize("tf", [True, False])
def test_window_s4_step_at_the_sentinel_edge(ora, gpu_device, tf):
    """DpCoreW::step<.., S4> gives even states a two-way maximum, so a dead state holds "< -1000" instead of exactly -1000;
    the window result is only used when the final score is above the sentinel.  Here every utterance takes the window +
    stride-4 step (T <= 1536, S <= 64, stride 4, no blank among the tokens) and its posteriors' planted peak is tuned so that the
    score of the best path lands within a few units of -1000, on both sides: whatever side it is on, states, tuples and
    mode must be the oracle's (and the utterances below the line go through the full-layout rerun)."""
    rng = np.random.default_rng(77 + int(tf))
    C, blank = 67, 66
    prm = ora.make_params(blank, 0, 10, True, tf)
    lps, toks, finals = [], [], []
    targets = np.concatenate([np.linspace(-1012.0, -990.0, 23), np.linspace(-1003.0, -999.0, 9)])
    for k, target in enumerate(targets):
        T, S = int(rng.integers(520, 1000)), int(rng.integers(36, 60))
        base, tk, planted = cases.planted_case(rng, T, S, C=C, blank=blank, peak=0.0, sigma=1.0)

        def scaled(peak):  # noise + peak * onehot(planted path)
            x = base.copy()
            x[np.arange(T), planted] += np.float32(peak)
            return torch.log_softmax(torch.from_numpy(x), dim=-1).numpy()

        def proxy(peak):  # score of the planted path on the prepared emissions: monotone in the peak
            _, mod = ora.prepare_emissions(scaled(peak), tk, prm)
            return float(mod[np.arange(T), planted].astype(np.float64).sum())
        lo, hi = 0.0, 12.0
        assert proxy(lo) < target < proxy(hi), (proxy(lo), proxy(hi))
        for _ in range(36):
            mid = 0.5 * (lo + hi)
            if proxy(mid) < target:
                lo = mid
            else:
                hi = mid
        lp = scaled(hi)
        _, mod = ora.prepare_emissions(lp, tk, prm)
        path, pidx, L = _stride4_path(tk, blank)
        _, _, _, _, fdp = ora.viterbi(mod, path, pidx, max(L // 4, 20), tf, blank)
        finals.append(float(fdp[L - 1]) if tf else float(fdp.max()))
        lps.append(lp)
        toks.append(tk)
    finals = np.array(finals)
    above = (finals > -1000.0) & (finals < -985.0)
    below = finals <= -1000.0
    assert above.sum() >= 6 and below.sum() >= 6, finals  # both sides of the sentinel, close to it
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10, tf=tf)
    _compare(res, exp, T_len)
    # the same with the stride-4 step disabled by construction (one target equal to the blank id is not allowed by the
    # reference's callers, so instead: the full layout via the window token limit) -- both must agree with the oracle
    res2, exp2 = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10, tf=tf, window_max_tokens=1)
    _compare(res2, exp2, T_len)


def _lp_with_sil_prob(p, C=67, filler=7):
    """log-"probabilities" whose boosted softmax (SIL = 0 is a target: +5) has P(SIL) = p[t]: column 0 = logit(p) - 5,
    one non-target filler column at 0, everything else far below"""
    p = np.asarray(p, np.float64)
    lp = np.full((len(p), C), -40.0, np.float32)
    lp[:, filler] = 0.0
    lp[:, 0] = (np.log(p) - np.log1p(-p) - 5.0).astype(np.float32)
    return lp


def _oscillating(n, centre, amp=0.005):
    """P(SIL) whose 10-frame sliding mean alternates (+,+,-,-) around `centre` with margin `amp`: a silent run every four
    frames (period 20 in the values)"""
    q = np.array([0.0, 0.05, 0.0, -0.05, 0.0, 0.05, 0.0, -0.05, 0.0, 0.0]) + centre + amp
    step = np.array([0.0, -0.1, 0.0, 0.1])
    p = np.empty(n)
    p[:10] = q
    for i in range(10, n):
        p[i] = p[i - 10] + step[(i - 10) % 4] * (amp / 0.005)
    return np.clip(p, 0.01, 0.995)


def test_planner_falls_back_to_the_global_scratch(ora, gpu_device):
    """The cooperative (LDS) planner holds 128 silence runs.  An utterance whose P(SIL) oscillates around the threshold has
    a run every four frames -- far more, although its T passes the planner's admission test.  The reference aligns such
    input; so must the device (by planning it again with the global scratch), not report BFA_ITEM_TOO_LARGE.  Case 2:
    the same for the sub-silences (threshold 0.8) of ONE speech piece between two real silences."""
    C, blank = 67, 66
    # 1) audio silences (threshold 0.9) overflow the LDS array
    p1 = _oscillating(1200, 0.9)
    # 2) two real silences, between them 700 frames whose mean oscillates around 0.8 (below 0.9: no audio silence there)
    p2 = np.concatenate([np.full(60, 0.05), np.full(60, 0.99), _oscillating(700, 0.8), np.full(60, 0.99), np.full(60, 0.05)])
    lps = [_lp_with_sil_prob(p1), _lp_with_sil_prob(p2)]
    toks = [np.array([0, 5, 9, 0, 11, 3, 0, 21, 0], np.int64), np.array([12, 0, 5, 9, 30, 31, 0, 14], np.int64)]
    prm = ora.make_params(blank, 0)
    for lp, tk, thr, lo in ((lps[0], toks[0], 0.9, 0), (lps[1], toks[1], 0.8, 120)):
        _, mod = ora.prepare_emissions(lp, tk, prm)
        runs = ora.detect_silence(mod[lo:lo + 1200], 0, thr, 10)
        assert len(runs) > 128, len(runs)  # the inputs do overflow the 128-entry LDS arrays
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10)
    assert (exp["status"] == 0).all()
    _compare(res, exp, T_len)


def test_staged_and_tuple_per_lane_post_stages_agree(ora, gpu_device):
    """bfa_postprocess / bfa_confidences run in k_postconf (probed probabilities staged in LDS) when the shapes fit its
    LDS, else in the tuple-per-lane kernels (k_postprocess, k_conf).  The same tuples through both -- the second time in
    a segment array padded to a capacity the staging cannot hold -- must give identical rows and confidence bits, and
    both must equal the oracle's chain; with and without the soft-boundary extension, softness 3 and 6."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch
    from bournemouth_forced_aligner_amd.utils import postprocess_batch
    rng = np.random.default_rng(2027)
    C = 67
    lp, tk, T_len, S_len = _mk_batch(rng, 24, C, (150, 700), (5, 60), peak=float(rng.choice([6.0, 3.0])), sigma=1.5,
                                     sil_rate=0.15, sil_len=(10, 40), repeat_rate=0.05)
    lpd = torch.from_numpy(lp).to(gpu_device)
    res = AlignmentUtils(C - 1, 0).decode_alignments_device(lpd, torch.from_numpy(tk), T_len, S_len)
    torch.cuda.synchronize()
    assert (res.status.cpu().numpy() == 0).all()
    cap = res.segs.shape[1]
    big = 2000  # 68 bytes of LDS per tuple (records + 2 K staged cells): > 60 KB, the staging kernel declines; <= 6500
    huge = 7000  # > 6500 slots: k_postprocess keeps the tuples in place and takes the means through LDS in chunks
    Sd = torch.from_numpy(np.asarray(S_len, np.int32))
    for extend, soft in ((True, 3), (True, 6), (False, 3)):
        s1, n1 = res.segs.clone(), res.seg_count.clone()
        s2 = torch.zeros((lp.shape[0], big, 4), dtype=torch.int32, device=gpu_device)
        s2[:, :cap] = res.segs
        n2 = res.seg_count.clone()
        s3 = torch.zeros((lp.shape[0], huge, 4), dtype=torch.int32, device=gpu_device)
        s3[:, :cap] = res.segs
        n3 = res.seg_count.clone()
        postprocess_batch(lpd, Sd, s1, n1, extend=extend, boundary_softness=soft)
        postprocess_batch(lpd, Sd, s2, n2, extend=extend, boundary_softness=soft)
        postprocess_batch(lpd, Sd, s3, n3, extend=extend, boundary_softness=soft)
        torch.cuda.synchronize()
        assert torch.equal(n1, n3)
        for b in range(lp.shape[0]):
            assert torch.equal(s1[b, :int(n1[b])], s3[b, :int(n3[b])]), f"in-place rows item {b} extend={extend}"
        c1, st1 = calculate_confidences_batch(lpd, s1, n1)
        c2, st2 = calculate_confidences_batch(lpd, s2, n2)
        torch.cuda.synchronize()
        assert torch.equal(n1, n2) and torch.equal(st1, st2)
        n = n1.cpu().numpy()
        a1, a2 = s1.cpu().numpy(), s2.cpu().numpy()
        b1, b2 = c1.cpu().numpy(), c2.cpu().numpy()
        segs0, cnt0 = res.segs.cpu().numpy(), res.seg_count.cpu().numpy()
        for b in range(lp.shape[0]):
            np.testing.assert_array_equal(a1[b, :n[b]], a2[b, :n[b]], err_msg=f"rows item {b} extend={extend}")
            assert (b1[b, :n[b]].view(np.int32) == b2[b, :n[b]].view(np.int32)).all(), f"confidences item {b}"
            tup = [tuple(int(v) for v in r) for r in segs0[b, :cnt0[b]]]
            cov = ora.ensure_target_coverage_default(tup, int(S_len[b]))
            ext = (ora.extend_soft_boundaries(lp[b], cov, soft) if extend else cov) if cov else []
            assert [tuple(int(v) for v in r) for r in a1[b, :n[b]]] == [tuple(e[:4]) for e in ext], f"oracle rows item {b}"
            if ext:
                rc, oc, _, _ = ora.confidences(lp[b], ext)
                assert rc == 0 and (b1[b, :n[b]].view(np.int32) == oc.view(np.int32)).all(), f"oracle confidences item {b}"


def test_postprocess_of_thousands_of_tuples(ora, gpu_device):
    """bfa_postprocess beyond 6 500 tuple slots per utterance (a path of thousands of tokens, or seg_cap = Tmax + 1 of a
    long recording): the tuples stay in the caller's array and the segment means go through LDS 2 048 tuples at a time,
    pass 2 of a chunk behind pass 1 of the next (bfa_post.hip: k_postprocess<.., BIG>).  Utterances of 1 .. 9 000 tuples
    (below one chunk, exactly two chunks, two chunks + 1, four and a half) whose neighbouring frames sit around both
    thresholds, tuples to drop among them (idx == -1, idx >= S), log-probs and raw logits: rows equal to the oracle's
    ensure_target_coverage + extend_soft_boundaries, softness 3 and 5."""
    from bournemouth_forced_aligner_amd.utils import postprocess_batch
    from bournemouth_forced_aligner_amd import _lib
    rng = np.random.default_rng(90210)
    C = 23
    counts = [1, 700, 4096, 4097, 9000, 2048]
    cap = 9300
    T = 4 * 9000 + 200
    B = len(counts)
    lp = np.log(np.clip(rng.random((B, T, C)).astype(np.float32) ** 8, 1e-7, 1.0)).astype(np.float32)
    lp -= np.log(np.exp(lp.astype(np.float64)).sum(-1, keepdims=True)).astype(np.float32)   # (roughly) normalised
    segs = np.zeros((B, cap, 4), np.int32)
    cnt = np.zeros(B, np.int32)
    S_len = np.zeros(B, np.int32)
    for b, n in enumerate(counts):
        t, rows = int(rng.integers(0, 30)), []
        for i in range(n):
            d = int(rng.integers(1, 4))
            ph = int(rng.integers(0, C - 1))
            idx = i if rng.random() > 0.03 else (-1 if rng.random() < 0.5 else n + 5)
            rows.append((ph, t, t + d, idx))
            lp[b, t:t + d, ph] = np.log(rng.random(d).astype(np.float32) * 0.9 + 0.05)
            # the frames around the tuple: probabilities around mean * 1e-3 and 10 ** -softness
            g = int(rng.integers(0, 4))
            lp[b, t + d:t + d + g, ph] = np.log((10.0 ** rng.uniform(-5.5, -2.0, g)).astype(np.float32))
            t += d + g
        segs[b, :n] = np.array(rows, np.int32)
        cnt[b] = n
        S_len[b] = n
    raw = lp + rng.normal(0, 3, (B, T, 1)).astype(np.float32)     # raw logits with the same log-softmax up to rounding
    from bournemouth_forced_aligner_amd import AlignmentUtils
    from bournemouth_forced_aligner_amd.forced_alignment import align_heads
    for use_raw in (False, True):
        if use_raw:
            # the row statistics (max, log-sum) a fused call leaves; the targets of that call do not matter here
            x = torch.from_numpy(raw).to(gpu_device)
            tk = torch.from_numpy(rng.integers(1, C - 1, (B, 12)).astype(np.int64))
            (_, stats), = align_heads([AlignmentUtils(C - 1, 0)], [x], [tk], [T] * B, [12] * B)
            ref_lp = torch.log_softmax(torch.from_numpy(raw), dim=-1).numpy()   # core.py:898-899
        else:
            x, stats, ref_lp = torch.from_numpy(lp).to(gpu_device), None, lp
        for soft in (3, 5):
            sd = torch.from_numpy(segs).to(gpu_device)
            nd = torch.from_numpy(cnt).to(gpu_device)
            postprocess_batch(x, torch.from_numpy(S_len), sd, nd, extend=True, boundary_softness=soft, row_stats=stats)
            torch.cuda.synchronize()
            got, gn = sd.cpu().numpy(), nd.cpu().numpy()
            for b in range(B):
                tup = [tuple(int(v) for v in r) for r in segs[b, :cnt[b]]]
                cov = ora.ensure_target_coverage_default(tup, int(S_len[b]))
                ext = ora.extend_soft_boundaries(ref_lp[b], cov, soft) if cov else []
                assert gn[b] == len(ext), (b, gn[b], len(ext))
                assert [tuple(int(v) for v in r) for r in got[b, :gn[b]]] == [tuple(e[:4]) for e in ext], f"item {b} soft {soft} raw {use_raw}"
            assert any(not np.array_equal(got[b, :gn[b]], np.array([r for r in segs[b, :cnt[b]] if 0 <= r[3] < S_len[b]], np.int32).reshape(-1, 4)) for b in range(B))  # something was extended


def test_one_kernel_small_batch_path(ora, gpu_device):
    """Small batches of ONE sliding-window class go through k_one (plan + window DP + full-layout rerun + walk in one
    kernel): peaky posteriors (window result used), flat posteriors (every utterance rerun with the full layout: the
    score ends at the sentinel) and a batch with an empty, a proportional and a too-short utterance among them, both
    head widths and both final-state rules, against the oracle; the same batches with the path switched off by the hint
    (no hint -> every class kernel) must give the same arrays."""
    for C in (67, 17):
        for peak, tf in ((9.0, True), (0.3, True), (0.5, False), (6.0, False)):
            rng = np.random.default_rng(int(peak * 10) + C + int(tf))
            lps, toks = [], []
            for k in range(20):
                T = int(rng.integers(560, 640)); S = int(rng.integers(18, 20))   # L = 73..77 -> Rw = 1, rerun class R = 2
                if k == 3: S = 0
                if k == 5: T, S = 19, 19                                           # stride 1 / proportional
                if k == 7: T, S = 9, 12                                            # audio too short
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=C - 1, peak=peak, sigma=1.0, repeat_rate=0.1)
                lps.append(lp); toks.append(tk)
            lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, C - 1)
            res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10, tf=tf)   # host lengths -> hint -> k_one
            _compare(res, exp, T_len)
            from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
            au = AlignmentUtils(C - 1, 0, silence_anchors=10, truly_forced=tf)
            hint = au.viterbi_decoder.class_mask_hint(T_len, S_len, False, n_classes=C)
            assert hint == (1 << 8) | _lib.HINT_NO_SILENCE_TARGETS     # what bfa_launch_align takes the one-kernel path on
            plain = au.viterbi_decoder.align_batch(torch.from_numpy(lp).to(gpu_device), torch.from_numpy(tk), T_len, S_len,
                                                   seg_cap=lp.shape[1] + 1, class_mask=None)   # no hint: the class kernels
            torch.cuda.synchronize()
            assert torch.equal(plain.status, res.status) and torch.equal(plain.seg_count, res.seg_count)
            ok = (res.status == 0)
            assert torch.equal(plain.frame_phonemes[ok], res.frame_phonemes[ok]) and torch.equal(plain.frame_phonemes_idx[ok], res.frame_phonemes_idx[ok])


def test_one_kernel_tile_and_group_edges(ora, gpu_device):
    """k_one's consumer takes the tiles in groups of two between barriers and looks at the give-up flag every fourth group:
    utterance lengths around every tile (16 frames) and group (32 frames) boundary and around the flag's period (128 frames),
    one length per call so that the call is ONE window class, each of the three classes (Rw = 1, 2, 3), sharp posteriors
    (the window's result stands) and flat ones (the consumer gives up on the way or ends at the sentinel: full-layout
    rerun in the same kernel), both final-state rules.  At least 90 of the 150 calls must be calls the library takes the one-kernel path on."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    C = 67
    taken = 0
    lens = [81, 95, 96, 97, 111, 112, 113, 127, 128, 129, 143, 144, 145, 159, 160, 161, 191, 192, 193, 255, 256, 257, 271, 272, 273, 383, 384, 385, 400, 511, 512, 513, 640, 641]
    for i, T in enumerate(lens):
        for S in (18, 22, 40, 57, 60, (T - 1) // 3 - 1):   # Rw = 1, 1, 2, 2, 3 (stride 4); a stride-3 path
            if 3 * S + 1 > T or S > 64:
                continue
            peak, tf = ((9.0, True), (0.3, True), (0.4, False), (7.0, False))[(i + S) % 4]
            rng = np.random.default_rng(7000 + 13 * T + S)
            lps, toks = [], []
            for k in range(5):
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=C - 1, peak=peak, sigma=1.0, repeat_rate=0.1)
                lps.append(lp); toks.append(tk)
            lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, C - 1)
            au = AlignmentUtils(C - 1, 0, silence_anchors=10, truly_forced=tf)
            hint = au.viterbi_decoder.class_mask_hint(T_len, S_len, False, n_classes=C)
            if hint is not None and (hint & ~(_lib.HINT_NO_SILENCE_TARGETS | _lib.HINT_UNIFORM_LENGTHS)) in (1 << 8, 1 << 9, 1 << 10):
                taken += 1
            res, exp = _run_both(ora, gpu_device, lp, tk, T_len, S_len, C, anchors=10, tf=tf)
            _compare(res, exp, T_len)
    assert taken >= 90, taken


def test_uniform_lengths_hint_only_moves_the_work(ora, gpu_device):
    """BFA_HINT_UNIFORM_LENGTHS lets each XCD take one contiguous eighth of the batch (workgroup id -> utterance slot through
    xcd_eighth): the same arrays as without the hint, for batch sizes that are and are not multiples of 8, in the window,
    the narrow full-layout and a split wide class; and against the oracle."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    C = 67
    for B, T, S in ((77, 600, 20), (64, 300, 30), (9, 400, 16), (67, 900, 100)):
        rng = np.random.default_rng(B * 1000 + T)
        lps, toks = [], []
        for k in range(B):
            lp, tk, _ = cases.planted_case(rng, T - int(rng.integers(0, T // 10)), S - int(rng.integers(0, 3)), C=C, blank=C - 1, peak=7.0, sigma=1.0)
            lps.append(lp); toks.append(tk)
        lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, C - 1)
        au = AlignmentUtils(C - 1, 0, silence_anchors=10)
        vd = au.viterbi_decoder
        hint = vd.class_mask_hint(T_len, S_len, False, n_classes=C) & ~_lib.HINT_UNIFORM_LENGTHS
        lpd, tkd = torch.from_numpy(lp).to(gpu_device), torch.from_numpy(tk)
        plain = vd.align_batch(lpd, tkd, T_len, S_len, seg_cap=lp.shape[1] + 1, class_mask=hint)
        moved = vd.align_batch(lpd, tkd, T_len, S_len, seg_cap=lp.shape[1] + 1, class_mask=hint | _lib.HINT_UNIFORM_LENGTHS)
        torch.cuda.synchronize()
        assert (plain.status.cpu().numpy() == 0).all()
        for name in ("status", "seg_count", "mode"):
            assert torch.equal(getattr(plain, name), getattr(moved, name)), (B, T, S, name)
        cnt = plain.seg_count.cpu().numpy()
        for b in range(B):  # (rows past seg_count and frames past T_len are unspecified)
            assert torch.equal(plain.segs[b, :cnt[b]], moved.segs[b, :cnt[b]]), (B, T, S, b)
            for name in ("frame_phonemes", "frame_phonemes_idx"):
                assert torch.equal(getattr(plain, name)[b, :T_len[b]], getattr(moved, name)[b, :T_len[b]]), (B, T, S, b, name)
        exp = ora.decode_alignments(lp, tk, T_len, S_len, ora.make_params(C - 1, 0, 10, True, True, True, True))
        _compare(moved, exp, T_len)


def _softness_args(**kw):
    import argparse
    d = dict(sigma=1.0, batch=128, frames=1000, tokens=40, tlo=300, thi=1870, tok_div=12, steps=1, warmup=1, parity=128)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("peak", [7.0, 3.0])
def test_soft_posteriors_standard_mode_every_utterance(gpu_device, peak):
    """OFF the planted peak-9 generator (tools/softness.py): logits N(0,1) + peak on the planted class.
    At peak 7 most 1000-frame utterances end below the reference's -1000 sentinel (forced_alignment.py:23,656-682), at peak 3
    all do: the fast window gives up (by extrapolation or when dead) and the exact window aligns the item, closed-form dead tail
    included.  Uniform call (the headline shape) and one mixed-length call (k_mix + the wide classes), every utterance against
    the oracle."""
    sys.path.insert(0, ROOT)
    from tools import softness
    for shape, kw in (("headline", dict(batch=192)), ("mixed", dict(batch=160))):
        a = _softness_args(**kw)
        a.parity = a.batch
        rec = softness.run_standard(a, shape, peak, gpu_device)
        assert rec["status_ok"], rec
        assert rec["parity"]["utterances"] >= a.batch * 0.9 and rec["parity"]["mismatching_utterances"] == 0, rec
        if shape == "headline":
            it = rec["items"]
            if peak == 3.0:   # every item through the exact window: after a fast window that gave up, or routed there at once
                assert it["exact_done"] in (0, a.batch) and it["exact_alive"] == 0 and rec["sample_share_at_sentinel"] == 1.0, rec
            else:   # (a call of this size is k_one's, which reruns its own items: no counters)
                assert rec["sample_share_at_sentinel"] > 0.2, rec


def test_soft_posteriors_real_text_chain(gpu_device):
    """The softest setting of the C5 proxy (bench.py --config c5proxy --peak 3): both heads from raw logits, mixed lengths up to
    1 870 frames (30 s), SIL in the targets.  Nothing is silence-anchored any more (every utterance takes its standard-mode
    fallback, most of them in the sentinel regime), every class of every posterior is above both soft-boundary thresholds, so
    every extension walk of core.py:717-805 runs to its limit -- k_postconf's wide windows.  Rows after coverage + soft
    boundaries bit-exact and confidences within 1e-4 against the oracle's whole chain, every utterance, both heads."""
    sys.path.insert(0, ROOT)
    from tools import softness
    for shape, peak, B in (("c5proxy", 3.0, 72), ("c5proxy", 6.0, 48), ("realtext", 5.0, 64)):
        a = _softness_args(batch=B)
        a.parity = 2 * B
        rec = softness.run_heads(a, shape, peak, gpu_device)
        assert rec["status_ok"], rec
        p = rec["parity"]
        assert p["utterances"] >= B * 0.9 and p["mismatching_utterances"] == 0 and p["confidence_beyond_1e-4"] == 0, rec


@pytest.mark.parametrize("peak", [9.0, 6.0, 3.0])
def test_full_batch_layout_of_a_silence_anchored_call_every_utterance(gpu_device, peak):
    """The launch layout of a FULL mixed-length silence-anchored batch (bench.py --config c5proxy), forced onto a call small
    enough to compare every utterance (BFA_OPT_WIDE_ANY_MAX_BATCH = 8, so that 64 utterances are a "full batch"): fallbacks of
    the window classes Rw <= 4 through k_mix, of Rw 6 / 8 through the exact-window class kernels, the slots no window fits in
    the pieces' launch, k_dp5_any by class set (R 12 / 16, then R 4 / 6 / 8), all auxiliary kernels of a head on one lane.
    Sharp (most utterances silence-anchored), middling and soft (none) posteriors; rows after coverage + soft boundaries bit-exact
    and confidences within 1e-4 against the oracle's whole chain, every utterance, both heads."""
    sys.path.insert(0, ROOT)
    from tools import softness
    from bournemouth_forced_aligner_amd import _lib
    h = _lib.handle(0, 0)
    _lib.check(_lib.lib().bfa_set_option(h, _lib.OPT_WIDE_ANY_MAX_BATCH, 8), h, "bfa_set_option")
    try:
        B = 64
        a = _softness_args(batch=B)
        a.parity = 2 * B
        rec = softness.run_heads(a, "c5proxy", peak, gpu_device)
    finally:
        _lib.check(_lib.lib().bfa_set_option(h, _lib.OPT_WIDE_ANY_MAX_BATCH, 512), h, "bfa_set_option")
    assert rec["status_ok"], rec
    p = rec["parity"]
    assert p["utterances"] >= B * 0.9 and p["mismatching_utterances"] == 0 and p["confidence_beyond_1e-4"] == 0, rec
    if peak == 9.0:
        assert min(rec["segmented_share"]) > 0.5, rec   # the anchored mode with its pieces
    if peak == 3.0:
        assert max(rec["segmented_share"]) == 0.0, rec  # every utterance a fallback


def test_reference_post_dp_methods_on_the_mirror_class(ora, gpu_device):
    """PhonemeTimestampAligner.ensure_target_coverage / extend_soft_boundaries_func (core.py:462, 682) called directly, the way
    a user of the reference class may: host lists in, host lists out, against the oracle's restatement of both."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    from bournemouth_forced_aligner_amd.core import PhonemeTimestampAligner
    rng = np.random.default_rng(77)
    C = 67
    lp, tk, T_len, S_len = _mk_batch(rng, 12, C, (120, 500), (5, 40), peak=5.0, sigma=1.2, sil_rate=0.1, sil_len=(10, 30))
    lpd = torch.from_numpy(lp).to(gpu_device)
    rows = AlignmentUtils(C - 1, 0).decode_alignments(lpd, torch.from_numpy(tk), T_len, S_len)
    al = PhonemeTimestampAligner(device=str(gpu_device))
    seqs = [tk[b, :S_len[b]].tolist() for b in range(len(rows))]
    cov = al.ensure_target_coverage(seqs, [list(r) for r in rows], seq_lens=S_len)
    ext = al.extend_soft_boundaries_func(lpd, cov, boundary_softness=3)
    for b in range(len(rows)):
        want_cov = ora.ensure_target_coverage_default([tuple(r) for r in rows[b]], int(S_len[b]))
        assert [tuple(r[:4]) for r in cov[b]] == [tuple(r[:4]) for r in want_cov] and all(r[4] is False for r in cov[b])
        want = ora.extend_soft_boundaries(lp[b], want_cov, 3) if want_cov else []
        assert [tuple(r[:4]) for r in ext[b]] == [tuple(e[:4]) for e in want], f"item {b}"
        assert all(len(r) == 5 for r in ext[b])
