"""-m gpu : the one-kernel path of mixed-length calls (bfa_dp4.inc: k_mix; bfa_kernels.hip: k_order) against the oracle.
A standard-mode call of 64 or more utterances whose lengths differ is aligned AND walked by one kernel, longest first:
the exact window (in closed form once the scores are dead) or the narrow full layout per workgroup, fills and error items
included; classes it has no body for (wide windows, wide full layouts, fast windows of stride-1/2 paths) keep their kernels
in the same call."""
import numpy as np
import pytest
import torch

import cases
from test_gpu_parity import _compare
from test_gpu_xwin import _dying, _run

pytestmark = pytest.mark.gpu


def _batch(rng, C, n, shapes, kinds=(9.0, 3.0, 0.3, "dying")):
    blank = C - 1
    lps, toks = [], []
    for k in range(n):
        T, S = shapes[k % len(shapes)]
        kind = kinds[(k // len(shapes)) % len(kinds)]
        peak = 9.0 if kind == "dying" else kind
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=1.0, repeat_rate=0.1)
        if kind == "dying" and T >= 8:
            lp = _dying(rng, lp, T)
        lps.append(lp)
        toks.append(tk)
    return cases.pad_batch(lps, toks, C, blank)


@pytest.mark.parametrize("C", [67, 17])
def test_mixed_call_every_item_kind(ora, gpu_device, C):
    """windows Rw 1..4 (stride 4 and 3), narrow full layouts (no band, stride 2 / 1), wide classes beside them, proportional
    fills, too-short errors, empty targets -- in one call, host hint and the library's own class choice, both final rules"""
    rng = np.random.default_rng(9100 + C)
    shapes = [(600, 20), (1000, 40), (81, 20), (90, 25), (1700, 68), (2400, 96), (300, 12), (61, 15), (64, 20),  # windows / no band
              (150, 60), (200, 110), (130, 100),                      # strides 2 / 1: fast window or full layout
              (900, 180), (400, 180), (1200, 250),                    # wide window, wide full layouts
              (40, 40), (50, 45),                                     # L > T with T >= S: proportional
              (10, 30), (0, 5), (7, 0), (1, 1), (2, 1), (5, 1), (16, 4), (17, 4), (32, 8), (33, 8)]  # errors, empty, tiny DPs
    lp, tk, T_len, S_len = _batch(rng, C, 4 * len(shapes), shapes)
    for tf in (True, False):
        for hint in (0, None):
            res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, C, tf, class_mask=hint)
            _compare(res, exp, T_len)


def test_mixed_call_dead_tails_continue_in_the_workgroup(ora, gpu_device):
    """BASELINE config 4's generator (T up to 3000, S = T // 25): the long utterances die on the way and their consumers go on
    in closed form (STOP = 3); 320 utterances so that workgroups are also dispatched behind retiring ones"""
    from tools import synth
    Tl, Sl = synth.c4_lengths(32768)
    by_len = np.argsort(Tl)
    pick = np.concatenate([by_len[-40:], by_len[::128][:256], by_len[:24]])
    lp, tk = synth.c4_utterances(pick, Tl[pick], Sl[pick], 67, 1004, "cpu")
    lp, tk = lp.numpy(), tk.numpy().astype(np.int64)
    for tf in (True, False):
        res, exp = _run(ora, gpu_device, lp, tk, Tl[pick], Sl[pick], 67, tf)
        _compare(res, exp, Tl[pick])


def test_mixed_call_flat_and_dying_long_paths(ora, gpu_device):
    """flat posteriors from frame 0 (dead within a few blocks) and posteriors that die half way, T up to 2600, every narrow
    window class, repeated tokens (states that cannot skip inside the path)"""
    rng = np.random.default_rng(9300)
    C = 67
    shapes = [(2600, 104), (2000, 80), (1500, 60), (1100, 44), (700, 28), (2210, 88), (1337, 53), (401, 100), (1800, 20)]
    lp, tk, T_len, S_len = _batch(rng, C, 72, shapes, kinds=(0.3, "dying", 2.0, 0.0))
    for tf in (True, False):
        res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, C, tf)
        _compare(res, exp, T_len)


def test_mixed_call_order_is_a_permutation(gpu_device):
    """every utterance of a large mixed call comes back aligned exactly once (k_order's counting sort covers every slot):
    statuses OK, tuple counts equal to the token counts on sharp posteriors"""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    from tools import synth
    B = 3000
    Tl, Sl = synth.c4_lengths(B)
    Tl = np.minimum(Tl, 900)
    Sl = np.maximum(1, Tl // 25)
    lp, tk = synth.c4_utterances(np.arange(B), Tl, Sl, 67, 77, gpu_device)
    au = AlignmentUtils(66, 0, silence_anchors=0)
    res = au.viterbi_decoder.align_batch(lp, tk, Tl, Sl, anchor_pauses=False)
    torch.cuda.synchronize()
    assert int((res.status != 0).sum()) == 0
    assert np.array_equal(res.seg_count.cpu().numpy(), Sl)


def test_mixed_call_of_long_paths_only(ora, gpu_device):
    """a mixed call whose utterances all have 60-420 tokens (the soak's "long" regime): window classes Rw 3..8 (exact, as
    there are more than 64 tokens), full layouts up to R = 16 -- k_mix takes the narrow ones, the class kernels the rest"""
    rng = np.random.default_rng(9400)
    C, blank = 67, 66
    lps, toks = [], []
    for _ in range(96):
        S = int(rng.integers(60, 420)); T = int(rng.integers(4 * S + 1, 4 * S + 900))
        if rng.integers(0, 4) == 0:
            T = int(rng.integers(3 * S + 1, 4 * S + 1))  # stride 3
        lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=float(rng.choice([9.0, 3.0, 0.2])), sigma=1.0, repeat_rate=0.1)
        lps.append(lp)
        toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    for tf in (True, False):
        for hint in (0, None):
            for max_tokens in (None, 4096):  # 4096: the wide classes Rw = 6 / 8 stay FAST windows beside k_mix's exact narrow ones
                res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, C, tf, max_tokens=max_tokens, class_mask=hint)
                st = res.status.cpu().numpy()
                bad = np.nonzero(st != exp["status"])[0]
                assert bad.size == 0, [(int(T_len[b]), int(S_len[b]), int(st[b])) for b in bad[:12]]
                _compare(res, exp, T_len)


def test_small_mixed_calls(ora, gpu_device):
    """calls of 2 ... 40 utterances of different lengths (the process_sentences_batch regime) take the mixed-length path too, with
    the consumers' band masks computed in lanes (k_mix<.., LM>, calls of <= 256 utterances): BASELINE config 4's generator
    incl. utterances that die on the way (closed form under the lane masks), flat posteriors, both final-state rules, host hint
    and the library's own class choice"""
    from tools import synth
    Tl, Sl = synth.c4_lengths(32768)
    by_len = np.argsort(Tl)
    for B in (2, 3, 7, 40):
        pick = np.concatenate([by_len[-(B // 2):], by_len[::(32768 // (B - B // 2))][:B - B // 2]])
        lp, tk = synth.c4_utterances(pick, Tl[pick], Sl[pick], 67, 1004, "cpu")
        lp, tk = lp.numpy(), tk.numpy().astype(np.int64)
        for tf in (True, False):
            for hint in (0, None):
                res, exp = _run(ora, gpu_device, lp, tk, Tl[pick], Sl[pick], 67, tf, class_mask=hint)
                _compare(res, exp, Tl[pick])
    rng = np.random.default_rng(9500)
    shapes = [(2600, 104), (700, 28), (1500, 60), (401, 100), (90, 25), (40, 40), (10, 30), (300, 12)]
    lp, tk, T_len, S_len = _batch(rng, 67, 16, shapes, kinds=(0.3, "dying", 9.0, 0.0))
    for tf in (True, False):
        res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, 67, tf)
        _compare(res, exp, T_len)
