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
