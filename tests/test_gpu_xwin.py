"""-m gpu : the EXACT window consumer (bfa_dp3.inc: DpCoreW<.., EX>) and the walk's side-band rule against the oracle.
An exact-window item computes the in-band states only, yet its result stands in every regime (above the sentinel, dying
on the way, dead from early on); tests/test_dead_tail.py proves the construction on the CPU, this file checks the HIP path."""
import numpy as np
import pytest
import torch

import cases
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _run(ora, dev, lp, tk, T_len, S_len, C, tf, max_frames=None, max_tokens=None, class_mask=0):
    from bournemouth_forced_aligner_amd import AlignmentUtils
    blank = C - 1
    au = AlignmentUtils(blank, 0, silence_anchors=0, truly_forced=tf)
    au.viterbi_decoder.window_max_frames = max_frames
    au.viterbi_decoder.window_max_tokens = max_tokens
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lp).to(dev), torch.from_numpy(tk), T_len, S_len,
                                         anchor_pauses=False, seg_cap=lp.shape[1] + 1, class_mask=class_mask)
    torch.cuda.synchronize()
    exp = ora.decode_alignments(lp, tk, T_len, S_len, ora.make_params(blank, 0, 0, True, tf))
    return res, exp


def _dying(rng, lp, T):
    """posteriors that are fine up to a random frame and flat from there on: the DP dies on the way"""
    t0 = int(rng.integers(T // 4, max(T // 4 + 1, 3 * T // 4)))
    C = lp.shape[1]
    flat = rng.normal(0.0, 0.3, size=(T - t0, C)).astype(np.float32) - np.float32(9.0)
    lp = lp.copy()
    lp[t0:] = torch.log_softmax(torch.from_numpy(flat), dim=-1).numpy() - np.float32(6.0)
    return lp


@pytest.mark.parametrize("C", [67, 17])
def test_exact_window_every_class_and_regime(ora, gpu_device, C):
    """every banded stride >= 3 item through the exact window (window_max_frames = 8 rules the fast window out): all six
    classes, T from L (pace 1) to T >> L, sharp / flat / dying posteriors, repeated tokens, both final-state rules"""
    rng = np.random.default_rng(8100 + C)
    blank = C - 1
    lps, toks = [], []
    for S in (15, 16, 21, 33, 40, 47, 48, 63, 64, 90, 121, 150, 187, 188, 220, 247):  # L = 4S+1 (or 3S+1): classes 1..8
        for T in (4 * S + 1, 4 * S + 2, 3 * S + 1, 3 * S + 7, 5 * S + 3, 9 * S + 11, 25 * S):
            for kind in range(4):
                peak = (9.0, 3.0, 0.3, 9.0)[kind]
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=peak, sigma=1.0, repeat_rate=0.15)
                if kind == 3:
                    lp = _dying(rng, lp, T)
                lps.append(lp)
                toks.append(tk)
    order = np.argsort([lp.shape[0] for lp in lps], kind="stable")
    for lo in range(0, len(order), 64):
        sel = order[lo:lo + 64]
        lp, tk, T_len, S_len = cases.pad_batch([lps[i] for i in sel], [toks[i] for i in sel], C, blank)
        for tf in (True, False):
            for hint in (0, None):  # the host hint (exact classes in bits 20-27) and the library's own choice
                res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, C, tf, max_frames=8, class_mask=hint)
                _compare(res, exp, T_len)


def test_fast_window_reruns_go_through_the_exact_window(ora, gpu_device):
    """default limits: flat / dying posteriors end the fast window at the sentinel; strides >= 3 are rerun by the exact
    window (Item::xw = XW_REDO), stride 2 by the full layout"""
    rng = np.random.default_rng(8200)
    C, blank = 67, 66
    lps, toks = [], []
    for S in (16, 25, 40, 47, 60, 64):
        for T in (4 * S + 1, 3 * S + 2, 2 * S + 5, 6 * S, 20 * S):
            for kind in range(3):
                lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=blank, peak=(0.3, 9.0, 2.0)[kind], sigma=1.0, repeat_rate=0.1)
                if kind == 1:
                    lp = _dying(rng, lp, T)
                lps.append(lp)
                toks.append(tk)
    lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, blank)
    for tf in (True, False):
        res, exp = _run(ora, gpu_device, lp, tk, T_len, S_len, C, tf)
        _compare(res, exp, T_len)


def test_exact_window_long_utterances(ora, gpu_device):
    """the mixed-length workload's long utterances (BASELINE config 4: T up to 3000, S = T // 25): beyond 1536 frames the
    exact window takes them; the longest end in the dead regime"""
    from tools import synth
    Tl, Sl = synth.c4_lengths(32768)
    pick = np.argsort(Tl)[[16500, 20000, 24000, 27000, 30000, 32000, 32700, 32767]]
    lp, tk = synth.c4_utterances(pick, Tl[pick], Sl[pick], 67, 1004, "cpu")
    lp, tk = lp.numpy(), tk.numpy().astype(np.int64)
    for tf in (True, False):
        res, exp = _run(ora, gpu_device, lp, tk, Tl[pick], Sl[pick], 67, tf)
        _compare(res, exp, Tl[pick])
