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


def test_process_sentence_plumbing(gpu_device):
    """Config C1 ('butterfly'): 75 frames, ph66 targets [29,10,58,9,43,56,23] through process_sentence with an
    injected posterior model and phonemiser -- the structure of the reference's result dict (core.py:1166-1179)."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    import cases
    rng = np.random.default_rng(1)
    toks = [29, 10, 58, 9, 43, 56, 23]
    T = 75
    planted = np.full(T, 66)
    for j, t in enumerate(toks):
        planted[5 + 9 * j: 5 + 9 * j + 6] = t
    logits = rng.normal(0, 1, (1, T, 67)).astype(np.float32)
    logits[0, np.arange(T), planted] += 8
    lg = rng.normal(0, 1, (1, T, 17)).astype(np.float32)
    lg[0, :, 16] += 3
    groups = [1, 2, 3, 4, 5, 6, 7]
    al = PhonemeTimestampAligner(posterior_fn=lambda w, wl: (torch.from_numpy(logits), torch.from_numpy(lg), [T]),
                                 phonemizer=lambda text: {"ph66": toks, "pg16": groups, "eipa": list("bʌɾɚflaɪ")[:7]},
                                 device="cuda:0")
    wav = torch.zeros(int(T * 268))
    res = al.process_sentence("butterfly", wav, do_groups=True)
    seg = res["segments"][0]
    assert [p["phoneme_id"] for p in seg["phoneme_ts"]] == toks
    for i, p in enumerate(seg["phoneme_ts"]):
        assert set(p) == {"phoneme_id", "phoneme_label", "ipa_label", "start_ms", "end_ms", "confidence",
                          "is_estimated", "target_seq_idx", "index"}
        assert p["end_ms"] >= p["start_ms"] and p["target_seq_idx"] == i and 0.0 < p["confidence"] <= 1.0
        assert abs(p["start_ms"] - (5 + 9 * i) * 16.75) < 3 * 16.75
    assert res == al.process_batch(["butterfly"], [wav], do_groups=True)[0]


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
