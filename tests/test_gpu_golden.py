"""-m gpu : the HIP path against golden vectors generated from the REFERENCE (tests/golden/)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hotpath_cases.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_hotpath_cases_against_reference(gold, gpu_device):
    from bournemouth_forced_aligner_amd import AlignmentUtils, _calculate_confidences
    meta = json.loads(str(gold["meta"]))
    for i, m in enumerate(meta):
        lp = torch.from_numpy(gold[f"c{i}_lp"]).to(gpu_device)
        tk = torch.from_numpy(gold[f"c{i}_tok"].astype(np.int64))
        au = AlignmentUtils(m["blank"], m["sil"], silence_anchors=m["anchors"], ignore_noise=m["ignore_noise"],
                            truly_forced=m["truly_forced"])
        args = (lp[None], tk[None] if tk.numel() else torch.zeros((1, 1), dtype=torch.int64), torch.tensor([m["T"]]),
                torch.tensor([m["S"]]))
        if m["error"]:
            with pytest.raises(ValueError, match="Audio too short to align"):
                au.decode_alignments(*args, boost_targets=m["boost"], enforce_minimum=m["enforce"])
            continue
        segs = au.decode_alignments(*args, boost_targets=m["boost"], enforce_minimum=m["enforce"])[0]
        np.testing.assert_array_equal(np.array(segs, np.int32).reshape(-1, 4), gold[f"c{i}_seg"], err_msg=f"case {i} {m}")
        if m["S"] > 0:
            fp, fi, score = au.viterbi_decoder.decode_with_forced_alignment(
                lp, tk, return_scores=True, boost_targets=m["boost"], enforce_minimum=m["enforce"],
                anchor_pauses=m["anchors"] > 0)
            np.testing.assert_array_equal(fp.cpu().numpy(), gold[f"c{i}_fph"], err_msg=f"case {i}")
            np.testing.assert_array_equal(fi.cpu().numpy(), gold[f"c{i}_fidx"], err_msg=f"case {i}")
            assert abs(score - m["score"]) <= 1e-6 * max(1.0, abs(m["score"]))
        if f"c{i}_simple" in gold.files:
            simp = au.decode_alignments_simple(*args)[0]
            np.testing.assert_array_equal(np.array(simp, np.int32).reshape(-1, 4), gold[f"c{i}_simple"], err_msg=f"simple {i}")
        if f"c{i}_conf" in gold.files:
            fs = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), False) for r in gold[f"c{i}_conf_in"]]
            cf = _calculate_confidences(lp, fs)
            got = np.array([c[5] for c in cf], np.float32)
            np.testing.assert_allclose(got, gold[f"c{i}_conf"], atol=1e-4, rtol=0)   # north-star tolerance
            np.testing.assert_allclose(got, gold[f"c{i}_conf"], atol=2e-7, rtol=0)   # what is actually achieved
            np.testing.assert_array_equal(np.array([[c[1], c[2]] for c in cf], np.int32), gold[f"c{i}_conf_se"])


def test_min_phoneme_prob_against_reference(gpu_device):
    """ViterbiDecoder.min_phoneme_prob other than 1e-8 (forced_alignment.py:16-20,70; bfa_params.min_log_prob): tuples,
    framewise states and the prepared emissions' bit patterns against the reference's own outputs
    (tests/golden/make_golden_minprob.py), C = 67 and 17, probabilities 1e-8 / 1e-4 / 1e-2 / 0.5 / 1e-20 / 0."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "minprob_cases.npz"))
    meta = json.loads(str(g["meta"]))
    for i, m in enumerate(meta):
        lp = torch.from_numpy(g[f"m{i}_lp"]).to(gpu_device)
        tk = torch.from_numpy(g[f"m{i}_tok"].astype(np.int64))
        au = AlignmentUtils(m["blank"], 0, silence_anchors=10, ignore_noise=True, truly_forced=m["truly_forced"])
        au.viterbi_decoder.min_phoneme_prob = m["prob"]   # (how a caller of the reference changes it, forced_alignment.py:850)
        segs = au.decode_alignments(lp[None], tk[None], torch.tensor([m["T"]]), torch.tensor([m["S"]]))[0]
        np.testing.assert_array_equal(np.array(segs, np.int32).reshape(-1, 4), g[f"m{i}_seg"], err_msg=f"case {i} {m}")
        fp, fi, _ = au.viterbi_decoder.decode_with_forced_alignment(lp, tk)
        np.testing.assert_array_equal(fp.cpu().numpy(), g[f"m{i}_fph"], err_msg=f"case {i}")
        np.testing.assert_array_equal(fi.cpu().numpy(), g[f"m{i}_fidx"], err_msg=f"case {i}")
        mod = au.viterbi_decoder.prepare_emissions(lp[None], tk[None], [m["T"]], [m["S"]])[0].cpu().numpy()
        assert (mod.view(np.int32) == g[f"m{i}_mod"].view(np.int32)).all(), f"case {i}: prepared emissions differ"


def test_narrow_widths_against_reference(gpu_device):
    """Posterior widths below one host vector (C = 3..15): the reference boosts and re-normalises rows of any width
    (forced_alignment.py:29-56), and torch sums fewer than sixteen exponentials one after the other.  log_softmax bit
    patterns, tuples and framewise states against the reference's outputs (tests/golden/make_golden_narrow.py)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, log_softmax
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "narrow_cases.npz"))
    for C in (3, 4, 5, 8, 11, 12, 15):
        got = log_softmax(torch.from_numpy(g[f"ls{C}_x"]).to(gpu_device)).cpu().numpy()
        assert (got.view(np.int32) == g[f"ls{C}_y"].view(np.int32)).all(), C
    meta = json.loads(str(g["meta"]))
    for i, m in enumerate(meta):
        lp = torch.from_numpy(g[f"n{i}_lp"]).to(gpu_device)
        tk = torch.from_numpy(g[f"n{i}_tok"].astype(np.int64))
        au = AlignmentUtils(m["blank"], 0, silence_anchors=0, ignore_noise=True, truly_forced=m["truly_forced"])
        segs = au.decode_alignments(lp[None], tk[None], torch.tensor([m["T"]]), torch.tensor([m["S"]]))[0]
        np.testing.assert_array_equal(np.array(segs, np.int32).reshape(-1, 4), g[f"n{i}_seg"], err_msg=f"case {i} {m}")
        fp, fi, _ = au.viterbi_decoder.decode_with_forced_alignment(lp, tk)
        np.testing.assert_array_equal(fp.cpu().numpy(), g[f"n{i}_fph"], err_msg=f"case {i}")
        np.testing.assert_array_equal(fi.cpu().numpy(), g[f"n{i}_fidx"], err_msg=f"case {i}")


def test_log_softmax_against_torch_bits(gold, gpu_device):
    from bournemouth_forced_aligner_amd import log_softmax
    for C in (67, 17):
        got = log_softmax(torch.from_numpy(gold[f"ls{C}_in"]).to(gpu_device)).cpu().numpy()
        assert (got.view(np.int32) == gold[f"ls{C}_out"].view(np.int32)).all()


def test_level2_pipeline_against_reference(gold, gpu_device):
    """extract_timestamps_from_segment_batch (core.py:811-964) with the acoustic model stubbed out: the 8-tuples
    (id, start_frame, end_frame, target_idx, is_estimated, confidence, start_ms, end_ms) of both heads."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    lc = torch.from_numpy(gold["l2_logits_class"])
    lg = torch.from_numpy(gold["l2_logits_group"])
    spec = gold["l2_spectral_lens"].tolist()
    B = lc.shape[0]
    seqs = [gold["l2_tokens"][b, :gold["l2_seq_lens"][b]].tolist() for b in range(B)]
    groups = [gold["l2_group_tokens"][b, :gold["l2_seq_lens"][b]].tolist() for b in range(B)]
    al = PhonemeTimestampAligner(posterior_fn=lambda w, wl: (lc, lg, spec), device="cuda:0")
    ts, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), gold["l2_wav_lens"].tolist(), seqs,
                                                        start_offset_times=[0.5 * b for b in range(B)],
                                                        group_sequences=groups, do_groups=True)
    for b in range(B):
        for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
            rows = ts[b][key]
            gi, gf = gold[f"l2_{short}{b}_int"], gold[f"l2_{short}{b}_flt"]
            got_i = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
            np.testing.assert_array_equal(got_i, gi, err_msg=f"{key} item {b}")
            got_f = np.array([[r[5], r[6], r[7]] for r in rows], np.float32).reshape(-1, 3)
            np.testing.assert_allclose(got_f[:, 0], gf[:, 0], atol=2e-7, rtol=0)
            np.testing.assert_array_equal(got_f[:, 1:], gf[:, 1:])
    # the same results as padded arrays (bulk consumers: no per-row Python objects)
    arr = al.extract_timestamps_from_logits(lc, lg, spec, seqs, gold["l2_wav_lens"].tolist(),
                                            start_offset_times=[0.5 * b for b in range(B)], group_sequences=groups,
                                            as_arrays=True)
    for key in ("phoneme_timestamps", "group_timestamps"):
        a = arr[key]
        for b in range(B):
            n = int(a["count"][b])
            rows = list(zip(*(a["rows"][b, :n, k].tolist() for k in range(4)), a["is_estimated"][b, :n].tolist(),
                            a["confidence"][b, :n].tolist(), a["start_ms"][b, :n].tolist(), a["end_ms"][b, :n].tolist()))
            assert rows == ts[b][key]


def test_simplified_pipeline_against_reference(gold, gpu_device):
    """extract_timestamps_from_segment_simplified (core.py:995-1044): log_softmax -> decode_alignments_simple ->
    convert_to_ms with a plain-int spectral length (float64 ms); expected rows from the reference itself
    (tests/golden/make_golden_host.py)."""
    import json
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    exp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_cases.json")))["simplified"]
    lc = torch.from_numpy(gold["l2_logits_class"])
    spec = gold["l2_spectral_lens"].tolist()
    B = lc.shape[0]
    seqs = [gold["l2_tokens"][b, :gold["l2_seq_lens"][b]].tolist() for b in range(B)]
    al = PhonemeTimestampAligner(posterior_fn=lambda w, wl: (lc, None, spec), device="cuda:0")
    ts = al.extract_timestamps_from_segment_simplified(torch.zeros(B, 16), gold["l2_wav_lens"].tolist(), seqs,
                                                       start_offset_times=[0.5 * b for b in range(B)], debug=False)
    for b in range(B):
        got = [[int(r[0]), int(r[1]), int(r[2]), int(r[3]), bool(r[4]), float(r[5]), float(r[6]), float(r[7])]
               for r in ts[b]["phoneme_timestamps"]]
        assert got == exp[b], f"item {b}"
    # tensor input: lengths are the counts of non-blank entries (core.py:1004-1005)
    tk = torch.from_numpy(gold["l2_tokens"].astype(np.int64))
    ts2 = al.extract_timestamps_from_segment_simplified(torch.zeros(B, 16), gold["l2_wav_lens"].tolist(), tk,
                                                        start_offset_times=[0.5 * b for b in range(B)], debug=False)
    assert [t["phoneme_timestamps"] for t in ts2] == [t["phoneme_timestamps"] for t in ts]


def test_level2_pipeline_with_completeness_against_reference(gpu_device):
    """ensure_completeness=True (core.py:516-657): stride-1 utterances whose best path skips targets; the missing
    targets come back as estimated rows, which then go through the soft-boundary and confidence passes."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l2_complete.npz"))
    lc, lg = torch.from_numpy(z["logits_class"]), torch.from_numpy(z["logits_group"])
    spec = z["spectral_lens"].tolist()
    B = lc.shape[0]
    seqs = [z["tokens"][b, :z["seq_lens"][b]].tolist() for b in range(B)]
    groups = [z["group_tokens"][b, :z["seq_lens"][b]].tolist() for b in range(B)]
    al = PhonemeTimestampAligner(posterior_fn=lambda w, wl: (lc, lg, spec), device="cuda:0", ensure_completeness=True)
    ts, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), z["wav_lens"].tolist(), seqs,
                                                        start_offset_times=0.25, group_sequences=groups, do_groups=True)
    n_est = 0
    for b in range(B):
        for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
            rows = ts[b][key]
            gi, gf = z[f"{short}{b}_int"], z[f"{short}{b}_flt"]
            got_i = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
            np.testing.assert_array_equal(got_i, gi, err_msg=f"{key} item {b}")
            got_f = np.array([[r[5], r[6], r[7]] for r in rows], np.float32).reshape(-1, 3)
            np.testing.assert_allclose(got_f[:, 0], gf[:, 0], atol=2e-7, rtol=0)
            np.testing.assert_array_equal(got_f[:, 1:], gf[:, 1:])
            n_est += int(got_i[:, 4].sum())
    assert n_est > 50


def _c1():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = json.load(open(os.path.join(here, "c1_butterfly.json"), encoding="utf-8"))
    z = np.load(os.path.join(here, "c1_butterfly.npz"))
    return d, torch.from_numpy(z["logits_class"]), torch.from_numpy(z["logits_group"])


def _c1_aligner(d, lc, lg, **kw):
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    T = d["T"]
    return PhonemeTimestampAligner(preset=None, device="cuda:0", posterior_fn=lambda w, wl: (lc, lg, [T] * lc.shape[0]),
                                   phonemizer=lambda text: dict(d["ts"]),
                                   phoneme_id_to_group_id=dict(zip(d["tokens"], d["groups"])),
                                   phoneme_id_to_label={int(k): v for k, v in d["phoneme_labels"].items()},
                                   group_id_to_label={int(k): v for k, v in d["group_labels"].items()}, **kw)


def _assert_segment_equal(got, exp):
    assert set(got) == set(exp), (sorted(got), sorted(exp))
    for key in exp:
        if key in ("phoneme_ts", "group_ts", "words_ts"):
            assert len(got[key]) == len(exp[key]), key
            for g, e in zip(got[key], exp[key]):
                assert set(g) == set(e)
                for f in e:
                    if f == "confidence":
                        assert abs(g[f] - e[f]) <= 1.2e-7, (key, f, g[f], e[f])
                    else:
                        assert g[f] == e[f], (key, f, g[f], e[f])
        elif key == "coverage_analysis":
            for f in exp[key]:
                if isinstance(exp[key][f], list):
                    assert sorted(got[key][f]) == sorted(exp[key][f])
                else:
                    assert got[key][f] == exp[key][f], f
        else:
            assert got[key] == exp[key], key


def test_c1_butterfly_process_sentence_against_reference(gpu_device):
    """BASELINE.json configs[0]: `process_sentence("butterfly", wav)` -- 75 frames, ph66 targets [29,10,58,9,43,56,23]
    -- compared with the result dict the REFERENCE's own process_sentence produced on the same logits with the same
    stubbed phonemiser / model (tests/golden/make_golden_l2.py): every key, ms bit-exact, confidences <= 1.2e-7."""
    d, lc, lg = _c1()
    al = _c1_aligner(d, lc, lg)
    wav = torch.zeros(1, d["wav_samples"])
    wav[0, ::7] = 0.1  # (the content only matters to the stubbed model; RMS normalisation must not divide by zero)
    res = al.process_sentence("butterfly", wav, do_groups=True)
    assert list(res) == ["segments"] and len(res["segments"]) == 1
    _assert_segment_equal(res["segments"][0], d["expected"]["segments"][0])
    # the reference's positional constructor order and its batch wrapper
    batch = al.process_sentences_batch(["butterfly "], [wav], do_groups=True)
    assert isinstance(batch, list) and len(batch) == 1
    _assert_segment_equal(batch[0]["segments"][0], d["expected"]["segments"][0])
    assert al.total_segments_processed == 2 and al.perfect_matches == 2
    # a 1-D waveform is taken as one channel; without do_groups there is no group_ts key (core.py:1194)
    seg = al.process_sentence("butterfly", wav[0])["segments"][0]
    assert "group_ts" not in seg and [p["phoneme_id"] for p in seg["phoneme_ts"]] == d["tokens"]


def test_process_segments_error_behaviour_matches_reference(gpu_device):
    """core.py:1348-1399: the 'Audio too short to align' ValueError propagates from the single-call branch and is
    swallowed (empty, fully-keyed segment results) only in the batched branch (batch_size < number of segments)."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    d, lc, lg = _c1()
    long_ts = dict(d["ts"], ph66=list(range(1, 61)) + list(range(1, 41)), pg16=[1] * 100)  # 100 phonemes > 75 frames
    T = d["T"]
    lc2, lg2 = lc.repeat(2, 1, 1), lg.repeat(2, 1, 1)

    def make(batch_rows):
        return PhonemeTimestampAligner(
            preset=None, device="cuda:0",
            posterior_fn=lambda w, wl: (lc2[:w.shape[0]], lg2[:w.shape[0]], [T] * w.shape[0]),
            phoneme_id_to_group_id={i: 1 for i in range(66)},
            phonemizer=lambda text: dict(long_ts) if text == "long" else dict(d["ts"]))
    wav = torch.full((1, d["wav_samples"]), 0.05)
    al = make(1)
    with pytest.raises(ValueError, match="Audio too short to align"):
        al.process_sentence("long", wav)
    srt = {"segments": [{"start": 0.0, "end": 1.25625, "text": "long"}, {"start": 0.0, "end": 1.25625, "text": "ok"}]}
    with pytest.raises(ValueError, match="Audio too short to align"):   # one call for both segments: propagates
        al.process_segments(srt, wav, batch_size=16)
    out = make(1).process_segments(srt, wav, batch_size=1)               # sliced: the failing slice becomes empty
    assert isinstance(out, list) and len(out) == 1 and len(out[0]["segments"]) == 2
    bad, good = out[0]["segments"]
    assert bad["phoneme_ts"] == [] and bad["words_ts"] == [] and bad["coverage_analysis"]["aligned_count"] == 0
    assert set(bad) >= {"coverage_analysis", "ipa", "word_num", "words", "phoneme_ts", "words_ts", "ph66", "pg16"}
    assert [p["phoneme_id"] for p in good["phoneme_ts"]] == d["tokens"]
    # segments shorter than 0.05 s are dropped by chop_wav (core.py:280-282); all dropped -> ValueError (core.py:1336)
    with pytest.raises(ValueError, match="audio chopping errors"):
        make(1).process_segments({"segments": [{"start": 0.0, "end": 0.01, "text": "ok"}]}, wav)


def test_level2_large_fixture_against_reference(gpu_device):
    """tests/golden/l2_large.npz: 64 utterances x 2 heads (2670 reference 8-tuples; boundary_softness 3 / 7 / 1, flat and
    sharp posteriors, silences) through the device pipeline: rows and ms bit-exact, confidences <= 1.2e-7."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l2_large.npz"))
    n = exact = 0
    for k in range(int(z["n_batches"])):
        pre = f"b{k}_"
        lc, lg = torch.from_numpy(z[pre + "logits_class"]), torch.from_numpy(z[pre + "logits_group"])
        spec = z[pre + "spectral_lens"].tolist()
        B = lc.shape[0]
        seqs = [z[pre + "tokens"][b, :z[pre + "seq_lens"][b]].tolist() for b in range(B)]
        groups = [z[pre + "group_tokens"][b, :z[pre + "seq_lens"][b]].tolist() for b in range(B)]
        al = PhonemeTimestampAligner(preset=None, posterior_fn=lambda w, wl: (lc, lg, spec), device="cuda:0",
                                     boundary_softness=int(z[pre + "softness"]))
        ts, _, _ = al.extract_timestamps_from_segment_batch(torch.zeros(B, 16), z[pre + "wav_lens"].tolist(), seqs,
                                                            start_offset_times=z[pre + "offsets"].tolist(),
                                                            group_sequences=groups, do_groups=True)
        for b in range(B):
            for key, short in (("phoneme_timestamps", "p"), ("group_timestamps", "g")):
                rows = ts[b][key]
                gi, gf = z[f"{pre}{short}{b}_int"], z[f"{pre}{short}{b}_flt"]
                got_i = np.array([[r[0], r[1], r[2], r[3], int(r[4])] for r in rows], np.int32).reshape(-1, 5)
                np.testing.assert_array_equal(got_i, gi, err_msg=f"batch {k} {key} item {b}")
                got_f = np.array([[r[5], r[6], r[7]] for r in rows], np.float32).reshape(-1, 3)
                np.testing.assert_allclose(got_f[:, 0], gf[:, 0], atol=1.2e-7, rtol=0)
                np.testing.assert_array_equal(got_f[:, 1:], gf[:, 1:])
                n += len(rows)
                exact += int((got_f[:, 0].copy().view(np.int32) == gf[:, 0].copy().view(np.int32)).sum())
    assert n == 2670 and exact > 0.9 * n, (n, exact)


def test_window_stitching_against_reference_and_oracle(gpu_device):
    """bfa_stitch_windows vs the reference's own stich_window_predictions outputs (golden, bit patterns), with plain
    and padded output rows, and vs the oracle on larger random shapes."""
    import os
    from oracle import oracle as ora
    from bournemouth_forced_aligner_amd import stich_window_predictions, stitch_total_frames
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stitch_cases.npz"))
    for k in range(int(z["n"])):
        x, y = z[f"c{k}_x"], z[f"c{k}_y"]
        alen, F, sr, wms, sms = (int(v) for v in z[f"c{k}_cfg"])
        for pad in (None, x.shape[3] + 5):
            got = stich_window_predictions(torch.from_numpy(x).to(gpu_device), alen, F, sr, wms, sms, row_stride=pad)
            assert tuple(got.shape) == y.shape
            assert (got.cpu().numpy().view(np.int32) == y.view(np.int32)).all(), f"case {k} pad {pad}"
    rng = np.random.default_rng(12)
    for B, NW, F, C in ((4, 61, 10, 67), (3, 124, 10, 17), (2, 33, 7, 67), (1, 2, 2, 40), (2, 9, 1, 5)):
        alen = 16000 * (160 + 80 * (NW - 1)) // 1000
        x = rng.normal(0, 4, size=(B, NW, F, C)).astype(np.float32)
        total = stitch_total_frames(alen, F)
        w = torch.cos(torch.linspace(-np.pi / 2, np.pi / 2, F)).numpy()
        rc, exp = ora.stitch_windows(x, w, total)
        if rc != 0:   # shapes the reference itself rejects: the library must reject them too
            with pytest.raises(RuntimeError):
                stich_window_predictions(torch.from_numpy(x).to(gpu_device), alen, F)
            continue
        got = stich_window_predictions(torch.from_numpy(x).to(gpu_device), alen, F).cpu().numpy()
        assert (got.view(np.int32) == exp.view(np.int32)).all(), (B, NW, F, C)


def test_integration_option_b_snippet(gold, gpu_device):
    """INTEGRATION.md, Option B: the binding a maintainer of the reference would paste into forced_alignment.py is
    extracted from the document and executed as it stands (pure ctypes, no import of this package), then reference
    golden cases go through its AlignmentUtils.decode_alignments -- tuples, the too-short ValueError, a bad token."""
    import re
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    sect = md[md.index("## Option B"):]
    code = re.search(r"```python\n(.*?)```", sect, re.S).group(1)
    driver = textwrap.dedent("""
        import json, sys, numpy as np, torch
        assert "bournemouth_forced_aligner_amd" not in sys.modules
        gold = np.load(sys.argv[1])
        meta = json.loads(str(gold["meta"]))
        out = []
        for i, m in enumerate(meta):
            if m["anchors"] != 10 or not m["boost"] or not m["enforce"]:
                continue
            lp = torch.from_numpy(gold[f"c{i}_lp"]).cuda()
            tk = torch.from_numpy(gold[f"c{i}_tok"].astype(np.int64))
            au = AlignmentUtils(m["blank"], m["sil"], silence_anchors=m["anchors"], ignore_noise=m["ignore_noise"],
                                truly_forced=m["truly_forced"])
            args = (lp[None], tk[None] if tk.numel() else torch.zeros((1, 1), dtype=torch.int64), torch.tensor([m["T"]]),
                    torch.tensor([m["S"]]))
            try:
                segs = au.decode_alignments(*args)[0]
                ok = (not m["error"]) and np.array_equal(np.array(segs, np.int32).reshape(-1, 4), gold[f"c{i}_seg"])
            except ValueError as e:
                ok = bool(m["error"]) and "Audio too short to align" in str(e)
            out.append(bool(ok))
        try:
            AlignmentUtils(66, 0).decode_alignments(torch.zeros(1, 50, 67).cuda(), torch.tensor([[3, 99, 4]]),
                                                    torch.tensor([50]), torch.tensor([3]))
            bad_token = False
        except IndexError:
            bad_token = True
        print(json.dumps({"cases": len(out), "ok": sum(out), "bad_token": bad_token,
                          "package_imported": "bournemouth_forced_aligner_amd" in sys.modules}))
    """)
    from bournemouth_forced_aligner_amd import _lib
    env = dict(os.environ, BFA_HIP_LIBRARY=_lib.SO_PATH)
    p = subprocess.run([sys.executable, "-c", code + "\n" + driver, GOLD], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["cases"] >= 3 and res["ok"] == res["cases"] and res["bad_token"] and not res["package_imported"], res


def test_heads_layout_option_does_not_change_results(gold, gpu_device):
    """bfa_set_option(BFA_OPT_CALLS_IN_FLIGHT): the two heads of a bfa_align_heads call on the pair of library streams
    (default) or head 0 on the caller's stream and the rest on the side stream -- same tuples, confidences and statistics."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    from bournemouth_forced_aligner_amd.forced_alignment import align_heads
    lc = torch.from_numpy(gold["l2_logits_class"]).to(gpu_device)
    lg = torch.from_numpy(gold["l2_logits_group"]).to(gpu_device)
    spec, slen = gold["l2_spectral_lens"].tolist(), gold["l2_seq_lens"].tolist()
    tk, tg = torch.from_numpy(gold["l2_tokens"]), torch.from_numpy(gold["l2_group_tokens"])
    aus = [AlignmentUtils(66, 0), AlignmentUtils(16, 0)]
    for au in aus:
        au.viterbi_decoder.handle_slot = 5
    outs = []
    for on in (False, True, False):
        _lib.set_calls_in_flight(gpu_device.index or 0, 5, on)
        res = align_heads(aus, [lc, lg], [tk, tg], spec, slen, post={"extend": True, "boundary_softness": 3, "confidences": True})
        torch.cuda.synchronize()
        outs.append([(r.segs.cpu().numpy().copy(), r.seg_count.cpu().numpy().copy(), r.conf.cpu().numpy().copy(), st.cpu().numpy().copy())
                     for r, st in res])
    for other in outs[1:]:
        for (s0, c0, f0, t0), (s1, c1, f1, t1) in zip(outs[0], other):
            keep = np.arange(s0.shape[1])[None, :] < c0[:, None]
            assert np.array_equal(c0, c1) and np.array_equal(s0[keep], s1[keep])
            assert np.array_equal(f0[keep].view(np.int32), f1[keep].view(np.int32))
            assert np.array_equal(np.nan_to_num(t0).view(np.int32), np.nan_to_num(t1).view(np.int32))


def test_fused_front_end_equals_two_pass(gpu_device):
    """raw logits -> bfa_align_heads (log_softmax inside K1's row preparation, row statistics for the
    sparse readers, both heads from one call) against the two-pass path (bfa_log_softmax, then one bfa_align_batch per
    head): every output array bitwise equal -- rows, counts, confidence bit patterns, ms -- on a mixed batch (silences
    -> segmented mode with anchored pieces, flat posteriors -> sentinel reruns, stride-1 / proportional utterances,
    one path beyond 1024 states)."""
    import cases
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    rng = np.random.default_rng(77)
    shapes = [(300, 40, 9.0, 0.0), (500, 70, 6.0, 0.2), (260, 60, 2.0, 0.0), (120, 100, 5.0, 0.0), (90, 90, 5.0, 0.1),
              (1400, 300, 7.0, 0.0), (700, 20, 1.0, 0.15), (64, 3, 8.0, 0.3), (1100, 120, 8.0, 0.1), (33, 30, 3.0, 0.0)]
    Tmax = max(s[0] for s in shapes)
    B = len(shapes)
    lc = np.zeros((B, Tmax, 67), np.float32)
    lg = rng.normal(0, 1, (B, Tmax, 17)).astype(np.float32)
    lc[:, :, 66] = 5.0
    seqs, spec, wl = [], [], []
    for b, (T, S, peak, silr) in enumerate(shapes):
        lp, tk, planted = cases.planted_case(rng, T, S, C=67, peak=peak, sil_rate=silr, sil_len=(10, 30), repeat_rate=0.05)
        lc[b, :T] = lp * np.float32(1.7) + np.float32(rng.normal(0, 3))  # logits: scaled and shifted log-probs
        pg = np.where(planted == 66, 16, np.where(planted == 0, 0, 1 + planted % 15))
        lg[b, np.arange(T), pg] += np.float32(peak * 0.7)
        seqs.append([int(x) for x in tk])
        spec.append(T)
        wl.append(T * 320)
    gmap = {p: (0 if p == 0 else 1 + p % 15) for p in range(66)}
    al = PhonemeTimestampAligner(preset=None, device="cuda:0", phoneme_id_to_group_id=gmap)
    a = al.extract_timestamps_from_logits(torch.from_numpy(lc), torch.from_numpy(lg), spec, seqs, wl, start_offset_times=0.5,
                                          as_arrays=True, fused=True)
    b2 = al.extract_timestamps_from_logits(torch.from_numpy(lc), torch.from_numpy(lg), spec, seqs, wl, start_offset_times=0.5,
                                           as_arrays=True, fused=False)
    for key in ("phoneme_timestamps", "group_timestamps"):
        assert (a[key]["count"] == b2[key]["count"]).all() and a[key]["count"].sum() > 300
        for f in ("rows", "is_estimated", "start_ms", "end_ms"):
            for i in range(B):
                n = int(a[key]["count"][i])
                np.testing.assert_array_equal(a[key][f][i, :n], b2[key][f][i, :n], err_msg=f"{key} {f} item {i}")
        for i in range(B):
            n = int(a[key]["count"][i])
            assert (a[key]["confidence"][i, :n].view(np.int32) == b2[key]["confidence"][i, :n].view(np.int32)).all()


def test_soft_boundary_mean_cases_against_reference(gpu_device):
    """bfa_postprocess (both kernels: probabilities staged in LDS / tuple-per-lane) on tests/golden/softmean_cases.npz: the
    segment mean of core.py:709-714 is torch's float32 cascade sum of a strided view, and the cases sit on the threshold it
    feeds (make_golden_softmean.py).  Every case alone and all cases of one family as one padded batch."""
    import json
    from bournemouth_forced_aligner_amd import postprocess_batch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "softmean_cases.npz"))
    meta = json.loads(str(g["meta"]))
    dev = gpu_device
    for cap in (4, 500):   # seg_cap 500 x T 460 does not fit k_postconf's LDS staging -> the tuple-per-lane kernel
        for k, m in enumerate(meta):
            lp = torch.from_numpy(g[f"s{k}_lp"]).to(dev)[None]
            tin = g[f"s{k}_in"]
            segs = torch.zeros((1, cap, 4), dtype=torch.int32, device=dev)
            segs[0, :len(tin)] = torch.from_numpy(tin).to(dev)
            cnt = torch.tensor([len(tin)], dtype=torch.int32, device=dev)
            postprocess_batch(lp, [int(tin[:, 3].max()) + 1], segs, cnt, extend=True, boundary_softness=3)
            n = int(cnt[0])
            np.testing.assert_array_equal(segs[0, :n].cpu().numpy(), g[f"s{k}_out"], err_msg=f"cap {cap} case {k} {m}")


def test_narrow_widths_from_raw_logits_against_reference(gpu_device):
    """bfa_align_heads with RAW LOGITS of fewer than sixteen columns (C = 2..15): F.log_softmax (core.py:898-899) fused into
    the alignment in torch's order for short rows (one exponential after the other), the row statistics it leaves, and the
    sparse readers on top of them (confidences, soft boundaries; rows nobody prepared get their statistics on demand in the
    same order).  Against the reference's outputs (tests/golden/make_golden_narrow_raw.py)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch, postprocess_batch
    from bournemouth_forced_aligner_amd.forced_alignment import align_heads
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "narrow_raw_cases.npz"))
    meta = json.loads(str(g["meta"]))
    assert {m["C"] for m in meta} == {2, 3, 4, 5, 8, 11, 12, 15}
    for i, m in enumerate(meta):
        T, S = m["T"], m["S"]
        pad = 9                                             # padded rows beyond the utterance: statistics on demand
        x = torch.zeros((1, T + pad, m["C"]), dtype=torch.float32)
        x[0, :T] = torch.from_numpy(g[f"r{i}_x"])
        x[0, T:] = torch.randn((pad, m["C"]), generator=torch.Generator().manual_seed(i))
        x = x.to(gpu_device)
        tk = torch.from_numpy(g[f"r{i}_tok"].astype(np.int64))[None]
        au = AlignmentUtils(m["blank"], 0, silence_anchors=0, ignore_noise=True, truly_forced=m["truly_forced"])
        (res, stats), = align_heads([au], [x], [tk], [T], [S])
        res.raise_for_status()
        n = int(res.seg_count[0])
        np.testing.assert_array_equal(res.segs[0, :n].cpu().numpy(), g[f"r{i}_seg"], err_msg=f"case {i} {m}")
        np.testing.assert_array_equal(res.frame_phonemes[0, :T].cpu().numpy(), g[f"r{i}_fph"], err_msg=f"case {i}")
        np.testing.assert_array_equal(res.frame_phonemes_idx[0, :T].cpu().numpy(), g[f"r{i}_fidx"], err_msg=f"case {i}")
        # the row statistics are log_softmax's: (x - max) - logsum reproduces torch's bits
        lsm = torch.log_softmax(x[0, :T].cpu(), dim=-1).numpy()
        st = stats[0, :T].cpu().numpy()
        mine = (x[0, :T].cpu().numpy() - st[:, :1]) - st[:, 1:2]
        assert (mine.view(np.int32) == lsm.view(np.int32)).all(), f"case {i}: row statistics"
        conf, cst = calculate_confidences_batch(x, res.segs, res.seg_count, row_stats=stats)
        assert int(cst[0]) == 0
        np.testing.assert_allclose(conf[0, :n].cpu().numpy(), g[f"r{i}_conf"], atol=2e-7, rtol=0, err_msg=f"case {i}")
        # soft boundaries over the PADDED rows, like the reference (core.py:705): the fixture was made without padding, so
        # the padded rows are given -30 logits on every target column (their probabilities stay below every threshold)
        x2 = x.clone()
        x2[0, T:] = 0.0
        x2[0, T:, :m["blank"]] = -30.0
        (res2, stats2), = align_heads([au], [x2], [tk], [T], [S])
        postprocess_batch(x2, [S], res2.segs, res2.seg_count, extend=True, boundary_softness=3, row_stats=stats2)
        n2 = int(res2.seg_count[0])
        ext = res2.segs[0, :n2].cpu().numpy()
        want = g[f"r{i}_ext"]
        # (a tuple that reaches the last real frame may extend into the padding only if the padding invites it: it does not)
        np.testing.assert_array_equal(ext, want, err_msg=f"case {i} {m}: soft boundaries")


def test_long_paths_against_reference(gpu_device):
    """CTC paths beyond 1 024 states run in the workgroup-wide kernel: eight waves of 16 states per lane up to 8 192 states,
    sixteen waves of 32 up to 32 768 (round 5; the reference has no limit, forced_alignment.py:181-192).  Tuples and framewise
    states against the reference's outputs (tests/golden/make_golden_long.py: L = 2 401, 8 401, 9 001, 12 001, 16 401), alone and
    as one batch with shorter neighbours; beyond 32 768 states the item is reported, not aligned."""
    from test_oracle_golden import _long_cases
    from bournemouth_forced_aligner_amd import AlignmentUtils
    from bournemouth_forced_aligner_amd import _lib
    batch = []
    for k, m, lp, tk, g in _long_cases():
        au = AlignmentUtils(66, 0, silence_anchors=0, ignore_noise=True, truly_forced=m["truly_forced"])
        lpd = torch.from_numpy(lp).to(gpu_device)
        res = au.decode_alignments_device(lpd[None], torch.from_numpy(tk.astype(np.int64))[None], [m["T"]], [m["S"]])
        res.raise_for_status()
        n = int(res.seg_count[0])
        np.testing.assert_array_equal(res.segs[0, :n].cpu().numpy(), g[f"g{k}_seg"], err_msg=str(m))
        np.testing.assert_array_equal(res.frame_phonemes[0, :m["T"]].cpu().numpy(), g[f"g{k}_fph"], err_msg=str(m))
        np.testing.assert_array_equal(res.frame_phonemes_idx[0, :m["T"]].cpu().numpy(), g[f"g{k}_fidx"], err_msg=str(m))
        if m["truly_forced"]:
            batch.append((m, lp, tk, g[f"g{k}_seg"]))
    # one call holding paths of both big kernels and short neighbours (every class of the launcher at once)
    import cases
    rng = np.random.default_rng(5)
    extra = [cases.planted_case(rng, T, S, C=67, peak=6.0)[:2] for (T, S) in ((300, 40), (1000, 90), (2000, 300))]
    items = [(lp, tk) for _m, lp, tk, _s in batch] + extra
    Tm, Sm = max(x[0].shape[0] for x in items), max(len(x[1]) for x in items)
    LP = np.zeros((len(items), Tm, 67), np.float32)
    TK = np.full((len(items), Sm), 66, np.int64)
    for i, (lp, tk) in enumerate(items):
        LP[i, :lp.shape[0]] = lp
        TK[i, :len(tk)] = tk
    au = AlignmentUtils(66, 0, silence_anchors=0, ignore_noise=True, truly_forced=True)
    res = au.decode_alignments_device(torch.from_numpy(LP).to(gpu_device), torch.from_numpy(TK), [x[0].shape[0] for x in items],
                                      [len(x[1]) for x in items])
    res.raise_for_status()
    for i, (_m, _lp, _tk, want) in enumerate(batch):
        np.testing.assert_array_equal(res.segs[i, :int(res.seg_count[i])].cpu().numpy(), want, err_msg=f"batched item {i}")
    # beyond the sixteen-wave kernel: 4 S + 1 > 32 768
    T, S = 33200, 8200
    lp = torch.zeros((1, T, 67), device=gpu_device)
    res = au.decode_alignments_device(lp, torch.ones((1, S), dtype=torch.int64), [T], [S])
    assert int(res.status[0]) == _lib.ITEM_TOO_LARGE
    with pytest.raises(RuntimeError):
        res.raise_for_status()
