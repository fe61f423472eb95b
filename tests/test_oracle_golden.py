"""Pins the CPU oracle (oracle/bfa_oracle.c) against golden vectors generated FROM THE REFERENCE
(tests/golden/make_golden.py).  Integer results and log_softmax bit patterns must be identical;
confidences go through torch.exp in the reference (MKL VML, not restatable) and are held to 2e-7."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hotpath_cases.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _params(ora, m):
    return ora.make_params(m["blank"], m["sil"], m["anchors"], m["ignore_noise"], m["truly_forced"], m["boost"],
                           m["enforce"])


def test_decode_alignments_matches_reference(ora, gold):
    meta = json.loads(str(gold["meta"]))
    n_seg_mode = n_err = 0
    for i, m in enumerate(meta):
        lp, tk = gold[f"c{i}_lp"], gold[f"c{i}_tok"]
        prm = _params(ora, m)
        res = ora.decode_alignments(lp[None], tk[None] if tk.size else np.zeros((1, 1), np.int32), [m["T"]], [m["S"]], prm)
        if m["error"]:
            assert res["status"][0] == ora.ERR_TOO_SHORT
            assert "Audio too short" in m["error"]
            n_err += 1
            continue
        assert res["status"][0] == 0, f"case {i}"
        got = np.array(ora.segments_as_lists(res)[0], np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(got, gold[f"c{i}_seg"], err_msg=f"case {i} {m}")
        if m["S"] > 0:
            np.testing.assert_array_equal(res["frame_ph"][0, :m["T"]], gold[f"c{i}_fph"], err_msg=f"case {i}")
            np.testing.assert_array_equal(res["frame_idx"][0, :m["T"]], gold[f"c{i}_fidx"], err_msg=f"case {i}")
            n_seg_mode += int(res["mode"][0] == ora.MODE_SEGMENTED)
            score = ora.alignment_score(lp, res["frame_ph"][0, :m["T"]])
            assert abs(score - m["score"]) <= 1e-9 * max(1.0, abs(m["score"]))
    assert n_seg_mode >= 6 and n_err == 1


def test_modified_log_probs_bit_exact(ora, gold):
    """boost + log_softmax + floor (forced_alignment.py:29-83) reproduce torch's float32 bits."""
    meta = json.loads(str(gold["meta"]))
    checked = 0
    for i, m in enumerate(meta):
        key = f"c{i}_mod"
        if key not in gold.files:
            continue
        rc, mod = ora.prepare_emissions(gold[f"c{i}_lp"], gold[f"c{i}_tok"], _params(ora, m))
        assert rc == 0
        assert (mod.view(np.int32) == gold[key].view(np.int32)).all(), f"case {i}"
        sil = ora.detect_silence(mod, 0, 0.9, max(m["anchors"], 1))
        np.testing.assert_array_equal(np.array(sil, np.int32).reshape(-1, 2), gold[f"c{i}_sil09"])
        checked += 1
    assert checked >= 20


def test_log_softmax_bit_exact(ora, gold):
    for C in (67, 17):
        got = ora.log_softmax_rows(gold[f"ls{C}_in"])
        assert (got.view(np.int32) == gold[f"ls{C}_out"].view(np.int32)).all()


def test_simple_mode_matches_reference(ora, gold):
    meta = json.loads(str(gold["meta"]))
    n = 0
    for i, m in enumerate(meta):
        key = f"c{i}_simple"
        if key not in gold.files:
            continue
        prm = ora.make_params(m["blank"], m["sil"], m["anchors"], m["ignore_noise"], m["truly_forced"], False, False)
        res = ora.decode_alignments(gold[f"c{i}_lp"][None], gold[f"c{i}_tok"][None], [m["T"]], [m["S"]], prm, simple=True)
        got = np.array(ora.segments_as_lists(res)[0], np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(got, gold[key], err_msg=f"case {i}")
        n += 1
    assert n >= 30


def test_viterbi_direct(ora, gold):
    vmeta = json.loads(str(gold["vmeta"]))
    for i, m in enumerate(vmeta):
        path = gold[f"v{i}_path"]
        rc, fph, fidx, _, _ = ora.viterbi(gold[f"v{i}_lp"], path, np.arange(m["L"]) - 1, m["bw"], m["truly_forced"], m["blank"])
        assert rc == 0
        np.testing.assert_array_equal(fph, gold[f"v{i}_fph"], err_msg=f"v{i} {m}")
        np.testing.assert_array_equal(fidx, gold[f"v{i}_fidx"], err_msg=f"v{i} {m}")


def test_confidences_and_ms(ora, gold):
    meta = json.loads(str(gold["meta"]))
    n = exact = 0
    for i, m in enumerate(meta):
        key = f"c{i}_conf"
        if key not in gold.files:
            continue
        fs = [tuple(r) for r in gold[f"c{i}_conf_in"]]
        rc, conf, st, en = ora.confidences(gold[f"c{i}_lp"], fs)
        assert rc == 0
        np.testing.assert_allclose(conf, gold[key], atol=2e-7, rtol=0)
        np.testing.assert_array_equal(np.stack([st, en], 1), gold[f"c{i}_conf_se"])
        n += len(fs)
        exact += int((conf == gold[key]).sum())
        segs = [(f[0], int(s), int(e), f[3]) for f, s, e in zip(fs, st, en)]
        a, b = ora.convert_to_ms(segs, m["T"], 0.25, m["T"] * 268, 16000)
        np.testing.assert_array_equal(np.stack([a, b], 1), gold[f"c{i}_ms"])
    assert n > 500 and exact > 0.8 * n


def test_level2_postprocessing(ora, gold):
    """core.py:925-956 on synthetic logits: ensure_target_coverage (default), extend_soft_boundaries_func,
    confidences, ms, sort -- the reference's 8-tuples are reproduced from the oracle's stages."""
    lc, lg = gold["l2_logits_class"], gold["l2_logits_group"]
    B = lc.shape[0]
    for head, logits, toks, blank in (("p", lc, gold["l2_tokens"], 66), ("g", lg, gold["l2_group_tokens"], 16)):
        lp = np.stack([ora.log_softmax_rows(logits[b]) for b in range(B)])
        prm = ora.make_params(blank, 0)
        res = ora.decode_alignments(lp, toks, gold["l2_spectral_lens"], gold["l2_seq_lens"], prm)
        assert (res["status"] == 0).all()
        for b in range(B):
            segs = ora.segments_as_lists(res)[b]
            segs = ora.ensure_target_coverage_default(segs, int(gold["l2_seq_lens"][b]))
            segs = ora.extend_soft_boundaries(lp[b], segs, 3)
            rc, conf, st, en = ora.confidences(lp[b], segs)
            assert rc == 0
            segs = [(s[0], int(a), int(e), s[3]) for s, a, e in zip(segs, st, en)]
            sm, em = ora.convert_to_ms(segs, int(gold["l2_spectral_lens"][b]), 0.5 * b, int(gold["l2_wav_lens"][b]), 16000)
            order = np.argsort(sm, kind="stable")
            gi, gf = gold[f"l2_{head}{b}_int"], gold[f"l2_{head}{b}_flt"]
            got_int = np.array([[segs[k][0], segs[k][1], segs[k][2], segs[k][3], 0] for k in order], np.int32).reshape(-1, 5)
            np.testing.assert_array_equal(got_int, gi, err_msg=f"head {head} item {b}")
            np.testing.assert_allclose(conf[order], gf[:, 0], atol=2e-7, rtol=0)
            np.testing.assert_array_equal(sm[order], gf[:, 1])
            np.testing.assert_array_equal(em[order], gf[:, 2])


def test_window_stitching_against_reference(ora):
    """stich_window_predictions (cupe2i/windowing.py:103-173): the oracle restatement reproduces the reference's own
    outputs bit for bit (golden vectors from tests/golden/make_golden_stitch.py); the cosine weights are the
    reference's torch.cos values."""
    import os
    import torch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stitch_cases.npz"))
    for k in range(int(z["n"])):
        x, y, w = z[f"c{k}_x"], z[f"c{k}_y"], z[f"c{k}_w"]
        alen, F, sr, wms, sms = (int(v) for v in z[f"c{k}_cfg"])
        total = ora.stitch_total_frames(alen, F, sr, wms, sms)
        assert total == y.shape[1]
        rc, got = ora.stitch_windows(x, w, total)
        assert rc == 0
        assert (got.view(np.int32) == y.view(np.int32)).all(), f"case {k}"
        # the weights this build computes itself are the reference's
        w2 = torch.cos(torch.linspace(-np.pi / 2, np.pi / 2, F)).numpy()
        assert (w2.view(np.int32) == w.view(np.int32)).all()


def oracle_level2(ora, logits, toks, seq_lens, spec, wav_lens, offsets, blank, softness):
    """The oracle's stages chained like core.py:897-956 for one head: list[B] of (int rows [n,5], conf, start_ms, end_ms)."""
    B = logits.shape[0]
    lp = np.stack([ora.log_softmax_rows(logits[b]) for b in range(B)])
    res = ora.decode_alignments(lp, toks, spec, seq_lens, ora.make_params(blank, 0))
    assert (res["status"] == 0).all()
    lists = ora.segments_as_lists(res)
    out = []
    for b in range(B):
        segs = ora.ensure_target_coverage_default(lists[b], int(seq_lens[b]))
        segs = ora.extend_soft_boundaries(lp[b], segs, int(softness))
        rc, conf, st, en = ora.confidences(lp[b], segs)
        assert rc == 0
        segs = [(s[0], int(a), int(e), s[3]) for s, a, e in zip(segs, st, en)]
        sm, em = ora.convert_to_ms(segs, int(spec[b]), float(offsets[b]), int(wav_lens[b]), 16000)
        order = np.argsort(sm, kind="stable")
        rows = np.array([[segs[k][0], segs[k][1], segs[k][2], segs[k][3], 0] for k in order], np.int32).reshape(-1, 5)
        out.append((rows, conf[order], sm[order], em[order]))
    return out


def test_level2_large_fixture(ora):
    """tests/golden/l2_large.npz (make_golden_l2.py): 64 utterances x 2 heads = 2670 reference 8-tuples, three
    boundary_softness settings, flat and sharp posteriors, silences."""
    z = np.load(os.path.join(os.path.dirname(GOLD), "l2_large.npz"))
    n = exact = 0
    for k in range(int(z["n_batches"])):
        pre = f"b{k}_"
        for head, logits, toks, blank in (("p", z[pre + "logits_class"], z[pre + "tokens"], 66),
                                          ("g", z[pre + "logits_group"], z[pre + "group_tokens"], 16)):
            got = oracle_level2(ora, logits, toks, z[pre + "seq_lens"], z[pre + "spectral_lens"], z[pre + "wav_lens"],
                                z[pre + "offsets"], blank, int(z[pre + "softness"]))
            for b, (rows, conf, sm, em) in enumerate(got):
                gi, gf = z[f"{pre}{head}{b}_int"], z[f"{pre}{head}{b}_flt"]
                np.testing.assert_array_equal(rows, gi, err_msg=f"batch {k} head {head} item {b}")
                np.testing.assert_allclose(conf, gf[:, 0], atol=1.2e-7, rtol=0)
                np.testing.assert_array_equal(sm, gf[:, 1])
                np.testing.assert_array_equal(em, gf[:, 2])
                n += len(conf)
                exact += int((conf.view(np.int32) == gf[:, 0].copy().view(np.int32)).sum())
    assert n == 2670 and exact > 0.9 * n, (n, exact)


def test_min_phoneme_prob_matches_reference(ora):
    """ViterbiDecoder.min_phoneme_prob other than 1e-8 (forced_alignment.py:16-20,70): the floor is the float32 logarithm
    torch computed (stored in the fixture); tuples, framewise states and the modified log-probs' bits must be the
    reference's (tests/golden/make_golden_minprob.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "minprob_cases.npz"))
    meta = json.loads(str(g["meta"]))
    floors = set()
    n_floor_matters = 0
    for i, m in enumerate(meta):
        lp, tk = g[f"m{i}_lp"], g[f"m{i}_tok"]
        minlog = float(g[f"m{i}_minlog"][0])
        floors.add(minlog)
        prm = ora.make_params(m["blank"], 0, 10, True, m["truly_forced"], True, True, min_log_prob=minlog)
        res = ora.decode_alignments(lp[None], tk[None], [m["T"]], [m["S"]], prm)
        assert res["status"][0] == 0
        got = np.array(ora.segments_as_lists(res)[0], np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(got, g[f"m{i}_seg"], err_msg=f"case {i} {m}")
        np.testing.assert_array_equal(res["frame_ph"][0, :m["T"]], g[f"m{i}_fph"], err_msg=f"case {i}")
        np.testing.assert_array_equal(res["frame_idx"][0, :m["T"]], g[f"m{i}_fidx"], err_msg=f"case {i}")
        rc, mod = ora.prepare_emissions(lp, tk, prm)
        assert rc == 0 and (mod.view(np.int32) == g[f"m{i}_mod"].view(np.int32)).all(), f"case {i}"
        _, mod_default = ora.prepare_emissions(lp, tk, ora.make_params(m["blank"], 0, 10, True, m["truly_forced"]))
        n_floor_matters += int(not np.array_equal(mod, mod_default))
    assert len(floors) == 6 and n_floor_matters >= 16  # the non-default floors do change the emissions


def test_narrow_widths_match_reference(ora):
    """Posterior widths below one host vector (C = 3..15; tests/golden/make_golden_narrow.py): torch sums the exponentials
    of such a row one after the other -- log_softmax bit patterns, boosted / floored emissions, framewise states and
    tuples are the reference's."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "narrow_cases.npz"))
    for C in (3, 4, 5, 8, 11, 12, 15):
        y = ora.log_softmax_rows(g[f"ls{C}_x"])
        assert (y.view(np.int32) == g[f"ls{C}_y"].view(np.int32)).all(), C
    meta = json.loads(str(g["meta"]))
    assert len(meta) == 28
    for i, m in enumerate(meta):
        lp, tk = g[f"n{i}_lp"], g[f"n{i}_tok"]
        prm = ora.make_params(m["blank"], 0, 0, True, m["truly_forced"])
        res = ora.decode_alignments(lp[None], tk[None], [m["T"]], [m["S"]], prm)
        got = np.array(ora.segments_as_lists(res)[0], np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(got, g[f"n{i}_seg"], err_msg=f"case {i} {m}")
        np.testing.assert_array_equal(res["frame_ph"][0, :m["T"]], g[f"n{i}_fph"], err_msg=f"case {i}")
        np.testing.assert_array_equal(res["frame_idx"][0, :m["T"]], g[f"n{i}_fidx"], err_msg=f"case {i}")
        rc, mod = ora.prepare_emissions(lp, tk, prm)
        assert rc == 0 and (mod.view(np.int32) == g[f"n{i}_mod"].view(np.int32)).all(), f"case {i}"


def test_soft_boundary_mean_cases(ora):
    """tests/golden/softmean_cases.npz (make_golden_softmean.py, from the reference): extend_soft_boundaries_func on tuples
    whose neighbouring frame sits BETWEEN the threshold torch's float32 cascade mean gives and the one a float64-accumulated
    mean would give, and on segments long enough for every level of the cascade (core.py:709-735)."""
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "softmean_cases.npz"))
    meta = json.loads(str(g["meta"]))
    assert sum(m["kind"] == "adversarial" for m in meta) >= 20 and sum(m["kind"] == "long" for m in meta) >= 10
    moved = 0
    for k, m in enumerate(meta):
        tin = [tuple(int(v) for v in r) for r in g[f"s{k}_in"]]
        got = np.array(ora.extend_soft_boundaries(g[f"s{k}_lp"], tin, 3), np.int32).reshape(-1, 4)
        np.testing.assert_array_equal(got, g[f"s{k}_out"], err_msg=f"case {k} {m}")
        moved += int(not np.array_equal(g[f"s{k}_in"], g[f"s{k}_out"]))
    assert moved >= 10  # (the cases are not all no-ops)


def _long_cases():
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "long_cases.npz"))
    meta = json.loads(str(g["meta"]))
    import cases
    for k, m in enumerate(meta):
        rng = np.random.default_rng(m["seed"])
        lp, tk, _ = cases.planted_case(rng, m["T"], m["S"], C=67, peak=m["peak"])
        assert np.array_equal(tk.astype(np.int32), g[f"g{k}_tok"]) and float(lp.astype(np.float64).sum()) == float(g[f"g{k}_lpsum"][0]), \
            "the regenerated input differs from the one the fixture was made from"
        yield k, m, lp, tk, g


def test_long_paths_match_reference(ora):
    """CTC paths of 2 401 .. 16 401 states (tests/golden/make_golden_long.py, from the reference; the reference has no limit,
    forced_alignment.py:181-192): strides 1 and 4, both final-state rules, one case whose scores reach the sentinel."""
    n = 0
    for k, m, lp, tk, g in _long_cases():
        res = ora.decode_alignments(lp[None], tk[None], [m["T"]], [m["S"]], ora.make_params(66, 0, 0, True, m["truly_forced"]))
        assert res["status"][0] == 0
        np.testing.assert_array_equal(np.array(ora.segments_as_lists(res)[0], np.int32).reshape(-1, 4), g[f"g{k}_seg"], err_msg=str(m))
        np.testing.assert_array_equal(res["frame_ph"][0][:m["T"]], g[f"g{k}_fph"])
        np.testing.assert_array_equal(res["frame_idx"][0][:m["T"]], g[f"g{k}_fidx"])
        n += 1
    assert n >= 5
