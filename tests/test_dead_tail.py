"""The dead sentinel regime of _viterbi_decode (forced_alignment.py:608-653) in closed form, proved against the oracle's
recurrence on the CPU: once every state of a DP is <= -1000 the scores of the following frames -- and with them the
backpointer codes -- are functions of frame-local data (oracle/bfa_oracle.c: ora_dead_tail_codes states the premises).
The HIP tail kernel (csrc/bfa_tail.inc) implements that closed form; this test is what allows it to."""
import numpy as np
import pytest

from oracle import oracle as O


def _path(seq, stride, blank):
    L = stride * len(seq) + 1
    path = np.full(L, blank, np.int32)
    pidx = np.full(L, -1, np.int32)
    path[1::stride][: len(seq)] = seq
    pidx[1::stride][: len(seq)] = np.arange(len(seq))
    return path, pidx


def _emissions(rng, T, C, kind):
    """log-probability-like rows (<= 0) that die quickly, seasoned with the values the closed form's roundings hinge on"""
    if kind == 0:    # flat: every column about -log C
        e = -np.abs(rng.normal(4.0, 1.0, (T, C)))
    elif kind == 1:  # wide range, incl. values past the ulp change at -1024
        e = -np.abs(rng.normal(0.0, 12.0, (T, C)))
    else:            # mostly tiny: f32(-1000 + e) == -1000 ties everywhere (|e| < 2**-15)
        e = -np.abs(rng.normal(0.0, 4e-5, (T, C)))
        e[rng.random((T, C)) < 0.3] -= rng.uniform(2.0, 9.0)
    e = e.astype(np.float32)
    m = rng.random((T, C))
    e[m < 0.03] = 0.0
    e[(m >= 0.03) & (m < 0.05)] = -0.0
    e[(m >= 0.05) & (m < 0.08)] = np.float32(-2.0 ** -15)          # exactly half an ulp of 1000: ties to even
    e[(m >= 0.08) & (m < 0.10)] = np.float32(-2.0 ** -15 * 1.0001)
    e[(m >= 0.10) & (m < 0.12)] = np.float32(-2.0 ** -14)
    e[(m >= 0.12) & (m < 0.14)] = np.float32(-24.0)                # -1024: the ulp doubles
    # the DP must start somewhere above the sentinel: frame 0 stays ordinary
    e[0] = -np.abs(rng.normal(4.0, 1.0, C)).astype(np.float32)
    if kind == 2 or rng.random() < 0.8:
        e[1:40] -= np.float32(30.0)  # ... and then die within ~35 frames (the others die of their own accord, or not at all)
    return e


def _case(rng, i):
    C = int(rng.choice([17, 67]))
    blank = C - 1
    stride = int(rng.choice([2, 4, 4, 4]))
    S = int(rng.integers(1, 40))
    seq = rng.integers(0, C - 1, S).astype(np.int32)  # (never the blank id)
    if rng.random() < 0.3 and S > 2:                 # repeated neighbours: can_skip differs for stride 2
        seq[1::2] = seq[0::2][: len(seq[1::2])]
    path, pidx = _path(seq, stride, blank)
    L = len(path)
    T = int(rng.integers(max(L, 60), max(L, 60) + 160))
    bw = max(L // 4, 20) if L > 60 else 0
    if rng.random() < 0.15:
        bw = max(L // 3, 30) if L > 60 else 0  # the segmented mode's band (forced_alignment.py:441)
    return T, C, blank, path, pidx, bw, _emissions(rng, T, C, i % 3)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_closed_form_equals_the_recurrence(seed):
    rng = np.random.default_rng(seed)
    n_tails = n_frames = 0
    for i in range(3000):
        T, C, blank, path, pidx, bw, lp = _case(rng, i)
        pace32 = bool(i % 7 == 0)  # decode_alignments_simple's float32 band arithmetic
        forced = bool(i % 2)
        rc, fph, fidx, st, fdp, K, t_dead = O.viterbi_trace(lp, path, pidx, bw, forced, blank, pace_f32=pace32)
        assert rc == O.OK
        if t_dead + 3 >= T:
            continue
        t_from = t_dead + 3  # dp[t_dead + 2] is the first frame in closed form; its successor's codes follow from it
        rc2, K2 = O.dead_tail_codes(lp, path, bw, t_from, pace_f32=pace32)
        assert rc2 == O.OK
        assert np.array_equal(K[t_from:], K2[t_from:]), (seed, i)
        # the final-state rule on dead scores: forced -> L - 1, free -> state 0 (the first of the exact -1000s)
        assert st[-1] == ((len(path) - 1) if forced else 0), (seed, i)
        assert np.all(fdp <= -1000.0) and fdp[0] == -1000.0
        n_tails += 1
        n_frames += T - t_from
    assert n_tails >= 2500 and n_frames > 100000  # 4 seeds: >= 10^4 dead tails


def test_structure_check_refuses_other_paths():
    rng = np.random.default_rng(5)
    C, blank = 17, 16
    lp = _emissions(rng, 80, C, 0)
    seq = rng.integers(0, C - 1, 12).astype(np.int32)
    for stride in (1, 3):  # consecutive states that can both skip
        path, _ = _path(seq, stride, blank)
        rc, _ = O.dead_tail_codes(lp, path, 0, 10)
        assert rc == O.ERR_ARG
    path, _ = _path(seq, 4, blank)
    lp[50, 3] = 0.25  # not a log-probability
    path[1] = 3
    rc, _ = O.dead_tail_codes(lp, path, 0, 10)
    assert rc == O.ERR_ARG


def test_c4_like_generator_dies_and_the_tail_matches():
    """planted-path posteriors of the bench's mixed-length workload (BASELINE config 4): boosting ~S target columns costs a
    blank frame ~0.5 nat, so the utterances beyond ~1 900 frames end in the dead regime"""
    pytest.importorskip("torch")
    from tools import synth
    Tl, Sl = synth.c4_lengths(32768)
    P = O.make_params(66, silence_id=0, silence_anchors=0)
    for g in (17937, 5650):  # T = 2594 / 2995
        T, S = int(Tl[g]), int(Sl[g])
        lp, toks = synth.c4_utterances([g], [T], [S], 67, 1004, "cpu")
        m = O.prepare_emissions(lp[0, :T].numpy(), toks[0, :S].numpy(), P)[1]
        path, pidx = _path(toks[0, :S].numpy(), 4, 66)
        L = len(path)
        bw = max(L // 4, 20)
        rc, fph, fidx, st, fdp, K, t_dead = O.viterbi_trace(m, path, pidx, bw, True, 66)
        assert rc == O.OK and 1500 < t_dead < T - 500, t_dead
        rc2, K2 = O.dead_tail_codes(m, path, bw, t_dead + 3)
        assert rc2 == O.OK and np.array_equal(K[t_dead + 3:], K2[t_dead + 3:])
        rw = O.win_class_for(L, bw)
        assert rw == 4
        rc3, K3, fdp3, margin = O.window_codes(m, path, bw, rw)
        assert rc3 == O.OK and np.array_equal(K, K3) and np.array_equal(fdp, fdp3) and margin >= 0


# ---- the sliding window + the closed form outside it == the full DP, in any regime -------------------------------------
def _wcase(rng, i):
    C = int(rng.choice([17, 67]))
    blank = C - 1
    stride = int(rng.choice([1, 2, 3, 4, 4, 4]))
    S = int(rng.integers(16, 160)) if stride >= 3 else int(rng.integers(31, 200))
    seq = rng.integers(0, C - 1, S).astype(np.int32)
    if rng.random() < 0.2:
        seq[rng.random(S) < 0.2] = blank  # targets equal to the blank id: other can_skip patterns
    if rng.random() < 0.3:
        seq[1::2] = seq[0::2][: len(seq[1::2])]
    path, pidx = _path(seq, stride, blank)
    L = len(path)
    T = int(L + rng.integers(0, 3 * L)) if rng.random() < 0.7 else L  # (L == T: pace exactly 1)
    kind = i % 4
    lp = _emissions(rng, T, C, kind % 3)
    if kind == 3:  # peaky: the planted path stays far above the sentinel
        lp = -np.abs(rng.normal(6.0, 2.0, (T, C))).astype(np.float32)
        seg = np.minimum((np.arange(T) * L) // T, L - 1)
        lp[np.arange(T), path[seg]] = -np.abs(rng.normal(0.0, 0.05, T)).astype(np.float32)
    return T, C, blank, path, pidx, lp


@pytest.mark.parametrize("seed", [21, 22])
def test_window_plus_closed_form_equals_the_full_dp(seed):
    rng = np.random.default_rng(seed)
    n = n_dead = 0
    margins = []
    for i in range(700):
        T, C, blank, path, pidx, lp = _wcase(rng, i)
        L = len(path)
        for bw in {max(L // 4, 20), max(L // 3, 30)} if L > 60 else ():
            rw = O.win_class_for(L, bw)
            if rw == 0:
                continue
            rc, fph, fidx, st, fdp, K, t_dead = O.viterbi_trace(lp, path, pidx, bw, bool(i % 2), blank)
            assert rc == O.OK
            for rw_try in {rw, min(8, rw + 1 if rw < 4 else 8)}:  # the narrowest class and a roomier one
                rc2, K2, fdp2, margin = O.window_codes(lp, path, bw, rw_try)
                assert rc2 == O.OK
                bad = np.argwhere(K != K2)
                assert bad.size == 0, (seed, i, bw, rw_try, bad[:4])
                assert np.array_equal(fdp.view(np.int32), fdp2.view(np.int32)), (seed, i, bw, rw_try)
                if rw_try == rw:
                    margins.append(margin)
            n += 1
            n_dead += int(t_dead < T)
    assert n > 600 and n_dead > 150 and n - n_dead > 150
    assert min(margins) >= 0  # the band's top (+2) never reaches the window's top
