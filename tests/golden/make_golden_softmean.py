#!/usr/bin/env python3
"""tests/golden/make_golden_softmean.py -- golden vectors that pin the SEGMENT MEAN of extend_soft_boundaries_func, from the
REFERENCE.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_softmean.py
Writes tests/golden/softmean_cases.npz.

core.py:709-714 takes `probs[start:end, phoneme].mean().item()` -- torch's float32 mean of a STRIDED view, which is ATen's
cascade sum (four interleaved accumulators, levels of sixteen rows), not a running sum -- and the strict passes compare
`probs[f, phoneme] >= min(mean * 1e-3, 1e-3)` (core.py:731-735, 751-755).  Each adversarial case here is built so that the
probability of the frame just outside a tuple lies BETWEEN the threshold torch's mean gives and the one a float64-accumulated
mean (what rounds 1-4 of this repository used) would give: a restatement with the wrong summation order moves that
boundary by one frame.  A second family holds long segments (up to 400 frames: every level of the cascade) with random
neighbours.  Stored: the log-probs, the tuples handed to the reference, and what the reference returned.
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refload  # noqa: E402

torch.set_num_threads(1)
C = 67


def ref_extend(al, lp, tuples, softness=3):
    fs = [[(int(p), int(s), int(e), int(i), False) for (p, s, e, i) in tuples]]
    out = al.extend_soft_boundaries_func(torch.from_numpy(lp)[None], fs, boundary_softness=softness)
    return np.array([[r[0], r[1], r[2], r[3]] for r in out[0]], np.int32).reshape(-1, 4)


def reachable_prob_between(lo, hi):
    """a float32 x with lo <= exp(x) < hi (as python floats of the float32 exponential), torch.exp, or None"""
    if not (0.0 < lo < hi):
        return None
    x0 = np.float32(math.log(lo))
    xs = [x0]
    for d in (np.float32(-np.inf), np.float32(np.inf)):
        x = x0
        for _ in range(6):
            x = np.nextafter(x, d)
            xs.append(x)
    xs = np.array(sorted(set(float(v) for v in xs)), np.float32)
    ps = torch.exp(torch.from_numpy(xs)).numpy()
    cr = np.exp(xs.astype(np.float64)).astype(np.float32)
    for x, p, q in zip(xs, ps, cr):
        if p == q and lo <= float(p) < hi:
            return np.float32(x)
    return None


def main():
    al = refload.core_aligner()
    rng = np.random.default_rng(20260930)
    out, meta = {}, []
    n_adv = 0
    trials = 0
    while n_adv < 24 and trials < 200000:
        trials += 1
        n = int(rng.integers(2, 90))
        gap = 14
        T = gap + n + gap
        ph = int(rng.integers(1, 66))
        lp = np.full((T, C), -30.0, np.float32)
        seg = np.log(np.clip(rng.random(n).astype(np.float32) ** 2, 1e-6, 1.0)).astype(np.float32)
        lp[gap:gap + n, ph] = seg
        probs = torch.exp(torch.from_numpy(lp))
        # torch.exp (MKL VML) differs from the correctly rounded exponential by 1 ulp on ~1 % of arguments: a case must not
        # depend on that (the restatements use the correctly rounded one)
        if not np.array_equal(probs[gap:gap + n, ph].numpy(), np.exp(seg.astype(np.float64)).astype(np.float32)):
            continue
        m_torch = probs[gap:gap + n, ph].mean().item()
        m_f64 = float(np.float32(probs[gap:gap + n, ph].double().sum().item() / n))
        if m_torch == m_f64 or min(m_torch, m_f64) >= 1.0:
            continue
        th_t, th_d = min(m_torch * 1e-3, 1e-3), min(m_f64 * 1e-3, 1e-3)
        x = reachable_prob_between(min(th_t, th_d), max(th_t, th_d))
        if x is None:
            continue
        side = int(rng.integers(0, 2))   # 0: the frame before the start (pass 1), 1: the frame at the end (pass 2, last tuple)
        if side == 0:
            lp[gap - 1, ph] = x
        else:
            lp[gap + n, ph] = x
        tuples = [(ph, gap, gap + n, 0)]
        got = ref_extend(al, lp, tuples)
        k = len(meta)
        out[f"s{k}_lp"] = lp
        out[f"s{k}_in"] = np.array(tuples, np.int32)
        out[f"s{k}_out"] = got
        meta.append(dict(kind="adversarial", side=side, n=n, mean_torch=m_torch, mean_f64=m_f64))
        n_adv += 1
    # long segments with neighbours: every level of the cascade (16 rows of 4 = 64 frames per level-0 chunk)
    for n in (63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 400):
        T = n + 60
        lp = (np.log(np.clip(rng.random((T, C)).astype(np.float32) ** 6, 1e-9, 1.0))).astype(np.float32)
        tuples, t, idx = [], 20, 0
        ph = int(rng.integers(1, 66))
        tuples.append((ph, t, t + n, idx))
        lp[t:t + n, ph] = np.log(np.clip(rng.random(n).astype(np.float32), 1e-4, 1.0)).astype(np.float32)
        lp[t - 12:t, ph] = np.log(np.float32(1e-3) * rng.random(12).astype(np.float32) * 0.9 + 1e-5).astype(np.float32)
        lp[t + n:t + n + 12, ph] = np.log(np.float32(1e-3) * rng.random(12).astype(np.float32) * 0.9 + 1e-5).astype(np.float32)
        got = ref_extend(al, lp, tuples)
        k = len(meta)
        out[f"s{k}_lp"] = lp
        out[f"s{k}_in"] = np.array(tuples, np.int32)
        out[f"s{k}_out"] = got
        meta.append(dict(kind="long", n=n))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "softmean_cases.npz"), **out)
    moved = sum(1 for k, m in enumerate(meta) if m["kind"] == "adversarial" and not np.array_equal(out[f"s{k}_in"], out[f"s{k}_out"]))
    print(f"wrote {len(meta)} cases ({n_adv} adversarial after {trials} trials, {moved} of them extended by the reference)")


if __name__ == "__main__":
    main()
