"""tests/cases.py -- small seeded input generators shared by the parity tests and the golden-vector
generator (inputs only; expected outputs always come from the reference or the oracle)."""
import numpy as np
import torch


def planted_case(rng, T, S, C=67, blank=None, sil=0, peak=9.0, sigma=1.0, sil_rate=0.0, sil_len=(12, 40),
                 repeat_rate=0.0):
    """Planted-path posteriors: tokens, a random monotone segmentation, logits = N(0,sigma) + peak*onehot,
    log_probs = torch.log_softmax(logits) (float32).  With sil_rate>0 some tokens are SIL and get long
    planted runs so that the silence-anchored (segmented) mode triggers."""
    if blank is None:
        blank = C - 1
    non_special = [c for c in range(C) if c != blank and c != sil]
    toks = rng.choice(non_special, size=S).astype(np.int64) if S > 0 else np.zeros(0, np.int64)
    for j in range(1, S):
        if rng.random() < repeat_rate:
            toks[j] = toks[j - 1]
    is_sil = rng.random(S) < sil_rate if S > 0 else np.zeros(0, bool)
    toks = np.where(is_sil, sil, toks)
    planted = np.full(T, blank, np.int64)
    if S > 0 and T > 0:
        # minimum durations: 2 for tokens (1 if tight), sil_len for SIL
        want = np.where(is_sil, rng.integers(sil_len[0], sil_len[1] + 1, size=S), rng.integers(2, 7, size=S))
        if want.sum() > T:
            want = np.maximum(1, (want * (T / (want.sum() + 1e-9))).astype(np.int64))
            while want.sum() > T and want.max() > 1:
                want[np.argmax(want)] -= 1
        slack = max(0, T - int(want.sum()))
        gaps = rng.multinomial(slack, np.ones(S + 1) / (S + 1)) if slack > 0 else np.zeros(S + 1, np.int64)
        t = 0
        for j in range(S):
            t += int(gaps[j])
            d = int(want[j])
            planted[t:min(T, t + d)] = toks[j]
            t += d
    logits = rng.normal(0.0, sigma, size=(T, C))
    if T > 0:
        logits[np.arange(T), planted] += peak
    lp = torch.log_softmax(torch.from_numpy(logits.astype(np.float32)), dim=-1).numpy()
    return lp, toks.astype(np.int64), planted


def pad_batch(lps, toks, C, blank, Tmax=None, Smax=None):
    B = len(lps)
    Tmax = Tmax or max(1, max(lp.shape[0] for lp in lps))
    Smax = Smax or max(1, max(len(t) for t in toks))
    # padded frames carry a flat (uniform) log-prob like a softmax of zeros
    out = np.full((B, Tmax, C), np.float32(-np.log(C)), np.float32)
    tk = np.full((B, Smax), blank, np.int64)
    T_len = np.zeros(B, np.int64)
    S_len = np.zeros(B, np.int64)
    for b in range(B):
        out[b, :lps[b].shape[0]] = lps[b]
        tk[b, :len(toks[b])] = toks[b]
        T_len[b] = lps[b].shape[0]
        S_len[b] = len(toks[b])
    return out, tk, T_len, S_len
