"""CPU-only tests: the C-ABI library loads and exports every symbol include/bfa.h declares, the host
logic of the Python mirror, the no-GPU failure mode, and the world_size-2 gather on gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "bfa.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bfa_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from bournemouth_forced_aligner_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "build it first: python __graft_entry__.py"
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bfa.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)
    # ... and NOTHING else: -fvisibility=hidden + the export map csrc/bfa_exports.map (~60 internal bfa_launch_* /
    # bfa_k1_* symbols were exported beside the ABI)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtDdBbWw"}
        assert exported == set(names), f"exported but not in bfa.h: {sorted(exported - set(names))}; missing: {sorted(set(names) - exported)}"
    lib.bfa_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.bfa_version()
    assert lib.bfa_abi_version() == 6


def test_params_default_and_workspace_query():
    from bournemouth_forced_aligner_amd import _lib
    lib = _lib.lib()
    p = _lib.BfaParams()
    lib.bfa_params_default(ctypes.byref(p), 66, 0)
    assert (p.blank_id, p.silence_id, p.silence_anchors, p.ignore_noise, p.truly_forced, p.boost_targets,
            p.enforce_minimum, p.simple, p.max_blanks) == (66, 0, 10, 1, 1, 1, 1, 0, 10)
    small = lib.bfa_workspace_bytes(256, 600, 20, 67, ctypes.byref(p))
    big = lib.bfa_workspace_bytes(4096, 1000, 40, 67, ctypes.byref(p))
    assert 0 < small < big < 2 * 1024 ** 3
    p.silence_anchors = 0
    assert lib.bfa_workspace_bytes(4096, 1000, 40, 67, ctypes.byref(p)) < big
    assert lib.bfa_workspace_bytes(0, 1000, 40, 67, ctypes.byref(p)) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the failure mode on a machine without a GPU")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    h = ctypes.c_void_p()
    assert _lib.lib().bfa_create(ctypes.byref(h), 0) == _lib.BFA_ERR_NO_DEVICE
    au = AlignmentUtils(66, 0)
    lp = torch.zeros((1, 10, 67))
    with pytest.raises(RuntimeError, match="no CPU implementation|needs an AMD GPU"):
        au.decode_alignments(lp, torch.tensor([[3, 4]]), torch.tensor([10]), torch.tensor([2]))


def test_argument_errors_match_reference():
    from bournemouth_forced_aligner_amd import AlignmentUtils
    au = AlignmentUtils(66, 0)
    with pytest.raises(ValueError, match="Phoneme sequences and lengths required for forced alignment"):
        au.decode_alignments(torch.zeros((1, 4, 67)))  # forced_alignment.py:878-879
    au.viterbi_decoder.set_blank_id(None)
    with pytest.raises(ValueError, match="Blank ID not set"):
        au.viterbi_decoder._params(True, True, True)  # forced_alignment.py:104-105


def test_assort_frames_host_rule(ora):
    from bournemouth_forced_aligner_amd import ViterbiDecoder
    rng = np.random.default_rng(3)
    for ign in (True, False):
        vd = ViterbiDecoder(66, 0, ignore_noise=ign)
        for _ in range(50):
            n = int(rng.integers(1, 80))
            ph = rng.choice([66, 66, 66, 5, 9], size=n)
            ix = np.where(ph == 66, -1, rng.integers(0, 3, size=n))
            reps = rng.integers(1, 14, size=n)
            ph, ix = np.repeat(ph, reps)[:200], np.repeat(ix, reps)[:200]
            got = vd.assort_frames(torch.from_numpy(ph), torch.from_numpy(ix))
            assert got == ora.assort_frames(ph, ix, 66, ign, 10)
    assert ViterbiDecoder(66, 0).assort_frames([], []) == []


def test_convert_to_ms_modes():
    from bournemouth_forced_aligner_amd import convert_to_ms
    fs = [(5, 10, 20, 0, False, 0.5), (7, 20, 31, 1, False, 0.25)]
    a = convert_to_ms(fs, 100, 0.5, 26800, 16000)               # python ints: float64 arithmetic
    assert a[0][6] == (0.5 + 10 * ((26800 / 16000) / 100)) * 1000
    b = convert_to_ms(fs, torch.tensor(100), 0.5, 26800, 16000)  # 0-dim tensor: float32 arithmetic (core.py:941)
    f32 = np.float32
    dpf = (f32(1) / f32(100)) * f32(26800 / 16000)
    assert b[1][7] == float((f32(0.5) + f32(31) * dpf) * f32(1000))
    assert len(convert_to_ms([(1, 2, 3)], 10, 0.0, 1600, 16000)[0]) == 8


def test_class_mask_hint():
    from bournemouth_forced_aligner_amd import ViterbiDecoder
    vd = ViterbiDecoder(66, 0, silence_anchors=10)
    NS = 1 << 16   # BFA_HINT_NO_SILENCE_TARGETS
    assert vd.class_mask_hint([1000] * 4, [40] * 4, has_sil=False) == NS | 0b10          # L=161 -> R=3
    assert vd.class_mask_hint([600], [20], has_sil=False) == NS | 0b1                     # L=81  -> R=2
    assert vd.class_mask_hint([1000], [40], has_sil=True) == 0b11                         # segments may be shorter
    assert vd.class_mask_hint([3000, 200], [120, 8], has_sil=False) == NS | 0b10001       # L=481 -> R=8, L=33 -> R=2
    # sliding-window classes (bits 8-15) on the 67-class head: L=161, bw=40 -> Rw=2; L=81, bw=20 -> Rw=1
    assert vd.class_mask_hint([1000] * 4, [40] * 4, has_sil=False, n_classes=67) == NS | 1 << 9
    assert vd.class_mask_hint([600], [20], has_sil=False, n_classes=67) == NS | 1 << 8
    assert vd.class_mask_hint([600], [20], has_sil=False, n_classes=40) == NS | 0b1       # other widths: generic kernel
    # more than 64 tokens: no window attempt (the path score would usually cross the sentinel) unless the limit is raised
    # -> the EXACT window of the same class (bits 20-27: Rw at bit 19+Rw), which stands in every regime
    assert vd.class_mask_hint([1500, 200], [120, 8], has_sil=False, n_classes=67) == NS | (1 << 23) | 0b1  # L=481 -> exact Rw=4
    assert vd.class_mask_hint([900], [180], has_sil=False, n_classes=67) == NS | 1 << 25     # L=721, bw=180 -> exact Rw=6
    assert vd.class_mask_hint([400], [180], has_sil=False, n_classes=67) == NS | 1 << 3      # stride 2 (L=361): no exact window, full layout R=6
    vd.window_max_tokens = 4096
    assert vd.class_mask_hint([1500], [120], has_sil=False, n_classes=67) == NS | (1 << 11)  # L=481 -> Rw=4
    assert vd.class_mask_hint([1500, 200], [120, 8], has_sil=False, n_classes=67) == NS | (1 << 23) | 0b1  # two lengths in one call: the mixed path, exact Rw=4
    assert vd.class_mask_hint([3000, 200], [120, 8], has_sil=False, n_classes=67) == NS | (1 << 23) | 0b1  # T > 1536: exact Rw=4
    assert vd.class_mask_hint([900], [180], has_sil=False, n_classes=67) == NS | 1 << 13     # L=721, bw=180 -> 372 states -> Rw=6
    # bit 17 (speed hint): 64 or more utterances with about the same number of frames
    from bournemouth_forced_aligner_amd._lib import HINT_UNIFORM_LENGTHS as UL
    assert vd.class_mask_hint([1000] * 64, [40] * 64, has_sil=False, n_classes=67) == NS | UL | 1 << 9
    assert vd.class_mask_hint([1000] * 63 + [900], [40] * 64, has_sil=False, n_classes=67) == NS | UL | 1 << 9
    # 64 or more utterances whose lengths differ: the mixed-length path (k_mix) -- the stride >= 3 window items of the classes
    # Rw <= 4 are exact-window items (bits 20-23); stride-2 paths keep the fast window.  Since the end of round 4 this holds
    # from TWO utterances on -- except a call of fewer than 64 utterances whose DPs all sit in ONE fast-window class (k_one's)
    assert vd.class_mask_hint([1000] * 63 + [500], [40] * 63 + [20], has_sil=False, n_classes=67) == NS | 1 << 21 | 1 << 20
    assert vd.class_mask_hint([1000] * 62 + [500], [40] * 62 + [20], has_sil=False, n_classes=67) == NS | 1 << 21 | 1 << 20
    assert vd.class_mask_hint([1000, 500], [40, 20], has_sil=False, n_classes=67) == NS | 1 << 21 | 1 << 20
    assert vd.class_mask_hint([1000, 700, 900], [40, 35, 42], has_sil=False, n_classes=67) == NS | 1 << 9   # one class (Rw=2): k_one
    assert vd.class_mask_hint([1000] * 63 + [300], [40] * 63 + [100], has_sil=False, n_classes=67) == NS | 1 << 21 | 1 << 9  # stride 2, L=201: fast Rw=2
    assert vd.class_mask_hint([1000] * 63 + [500], [40] * 63 + [20], has_sil=True, n_classes=67) == 0b11 | 1 << 9 | 1 << 8  # SIL possible: no mixed path (the library drops the window bits in that mode)
    assert vd.class_mask_hint([900], [187], has_sil=False, n_classes=67) == NS | 1 << 15     # L=749, bw=187 -> 390 states -> Rw=8
    assert vd.class_mask_hint([1200], [250], has_sil=False, n_classes=67) == NS | 1 << 6     # L=1001, bw=250: 516 states > 512 -> full R=16
    assert vd.class_mask_hint([], [], has_sil=False) == 0


def test_lpt_sharding_is_a_partition_and_balanced():
    from bournemouth_forced_aligner_amd.sharding import shard_utterances, utterance_cost
    rng = np.random.default_rng(0)
    T = rng.integers(200, 3001, size=4096)
    S = np.maximum(1, T // 25)
    shards = shard_utterances(T, S, 8)
    allidx = np.sort(np.concatenate(shards))
    np.testing.assert_array_equal(allidx, np.arange(4096))
    loads = np.array([utterance_cost(T[s], S[s]).sum() for s in shards], np.float64)
    assert loads.max() / loads.mean() < 1.01


def _gloo_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bournemouth_forced_aligner_amd.sharding import gather_results, shard_utterances
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    n = 37
    T = rng.integers(50, 400, size=n)
    S = np.maximum(1, T // 25)
    cap = int(S.max()) + 2
    segs = rng.integers(0, 500, size=(n, cap, 4)).astype(np.int32)
    cnt = rng.integers(0, cap + 1, size=n).astype(np.int32)
    conf = rng.random((n, cap)).astype(np.float32)
    shards = shard_utterances(T, S, world)
    mine = shards[rank]
    local_cap = int(S[mine].max()) + 2  # ranks may use different capacities
    out = gather_results(torch.from_numpy(segs[mine][:, :local_cap]), torch.from_numpy(np.minimum(cnt[mine], local_cap)),
                         torch.from_numpy(conf[mine][:, :local_cap]), torch.from_numpy(mine), n, dst=0)
    if rank == 0:
        gs, gc, gf = out.to_padded(cap)   # the padded arrays one call would have returned, original order
        ok = not out.overflowed() and out.records.shape[0] == world
        lists = out.to_lists()
        for i in range(n):
            r = next(k for k in range(world) if i in shards[k])
            lc = int(S[shards[r]].max()) + 2
            c = min(int(cnt[i]), lc)
            ok &= int(gc[i]) == c
            ok &= bool((gs[i, :c].numpy() == segs[i, :c]).all()) and bool((gs[i, c:] == 0).all())
            ok &= bool((gf[i, :c].numpy() == conf[i, :c]).all())
            t, cf = out.rows(i)             # straight from the packed records
            ok &= t.shape == (c, 4) and bool((t == segs[i, :c]).all()) and bool((cf == conf[i, :c]).all())
            ok &= lists[i] == [tuple(x) for x in segs[i, :c].tolist()]
        open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gather_on_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "ok")).read() == "1"


def test_textgrid_writer_matches_the_reference_examples():
    """dict_to_textgrid (utils.py:152-411) against the reference's own example pairs (data files) and the
    with-confidence / missing-tier variants its writer produces (tests/golden/make_golden_textgrid.py)."""
    import json
    import os
    from bournemouth_forced_aligner_amd.textgrid import dict_to_textgrid, dict_to_textgrid_with_confidence
    here = os.path.join(os.path.dirname(__file__), "golden", "textgrid")
    exp = json.load(open(os.path.join(here, "expected_variants.json"), encoding="utf-8"))
    for stem in ("LJ001-0001", "LJ001-0002"):
        d = json.load(open(os.path.join(here, stem + ".vs.json"), encoding="utf-8"))
        assert dict_to_textgrid(d) == open(os.path.join(here, stem + ".TextGrid"), encoding="utf-8").read()
        assert dict_to_textgrid(d, include_confidence=True) == exp[stem]
        assert dict_to_textgrid_with_confidence(d) == exp[stem]
        d2 = {"segments": [dict(d["segments"][0])]}
        d2["segments"][0].pop("words_ts", None)
        assert dict_to_textgrid(d2) == exp[stem + ":no_words"]
        assert dict_to_textgrid(d2, include_confidence=True) == exp[stem + ":no_words:conf"]
    assert dict_to_textgrid({"segments": [{"start": 0.0, "end": 2.5}]}) == exp["empty"]
    with pytest.raises(ValueError):
        dict_to_textgrid({"segments": []})


def test_backtrace_kernel_keeps_four_waves_per_simd():
    """A 4096-utterance batch is only resident at once in K2 if the NARROW walk kernel (window Rw <= 4, full layout
    R <= 4) fits 4 waves per SIMD (<= 128 VGPRs); every new per-layout instantiation of the walk counts against that
    (the compiler's resource remark is the check).  The wide kernel (R >= 6, Rw 6 / 8) trades occupancy for registers."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bournemouth-forced-aligner_amd", "csrc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                          "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "-c",
                          os.path.join(csrc, "bfa_backtrace.hip"), "-o", os.devnull],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {m.group(1): (int(m.group(2)), int(m.group(3)))
           for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                                out.stderr, re.S)}
    narrow = [v for k, v in res.items() if "k_backtraceE" in k]
    wide = [v for k, v in res.items() if "k_backtrace_wide" in k]
    assert narrow and wide, sorted(res)
    assert narrow[0][1] >= 4 and narrow[0][0] <= 72, res   # (13 dwords spill in the per-item prologue, none inside the step loops)
    assert wide[0][1] >= 2 and wide[0][0] <= 32, res       # long utterances, few of them: may use more registers


# ---- host-side result shaping against the reference's outputs (tests/golden/host_cases.json) ----
@pytest.fixture(scope="module")
def host_gold():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_cases.json")))


def _host_aligner(**kw):
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    return PhonemeTimestampAligner(device="cpu", **kw)  # no GPU work is started by these helpers


def test_align_words_matches_reference(host_gold):
    import copy
    al = _host_aligner()
    for c in host_gold["words"]:
        assert al._align_words(copy.deepcopy(c["phoneme_ts"]), c["word_num"], c["words"]) == c["expected"]


def test_coverage_analysis_and_frame_runs_match_reference(host_gold):
    al = _host_aligner()
    labels = {i: f"L{i}" for i in range(0, 66, 2)}
    for c in host_gold["coverage"]:
        tgt = torch.tensor(c["target"]) if c["tensor"] else c["target"]
        got = al.analyze_alignment_coverage(tgt, [tuple(r) for r in c["aligned"]], labels)
        got["missing_phonemes"], got["extra_phonemes"] = sorted(got["missing_phonemes"]), sorted(got["extra_phonemes"])
        assert got == c["expected"]
    for c in host_gold["compress"]:
        runs = al.compress_frames(c["frames"])
        assert [list(r) for r in runs] == c["expected"]
        assert al.decompress_frames(runs) == c["frames"]
    assert al.compress_frames([]) == []
    assert [al.ceil(x) for x in (2.0, 2.25, -2.5, -3.0, 0.0)] == [2, 3, -1, -3, 0]


def test_framewise_assortment_matches_reference(host_gold):
    import copy
    al = _host_aligner()
    for i, c in enumerate(host_gold["framewise"]):
        ts = copy.deepcopy(c["ts"])
        got = al.framewise_assortment(ts, c["total_frames"], c["fps"], gap_contraction=c["gap_contraction"],
                                      select_key=c["key"], offset_ms=c["offset_ms"])
        assert got == c["expected"], f"case {i}"
        assert [t["start_ms"] for t in ts] == sorted(t["start_ms"] for t in c["ts"])  # sorted in place
    with pytest.raises(ValueError):
        al.framewise_assortment([{"start_ms": 0.0, "end_ms": 10.0}], 10, 100.0)


def test_post_process_segment_matches_reference(host_gold):
    al = _host_aligner(phoneme_id_to_label={i: f"P{i}" for i in range(60)}, group_id_to_label={i: f"G{i}" for i in range(14)})
    for c in host_gold["post"]:
        got = al.post_process_segment(dict(c["segment"]), dict(c["ts"]), torch.tensor(c["seq"]),
                                      [tuple(r) for r in c["rows"]],
                                      [tuple(r) for r in c["grows"]] if c["grows"] is not None else None)
        ca = got["coverage_analysis"]
        ca["missing_phonemes"], ca["extra_phonemes"] = sorted(ca["missing_phonemes"]), sorted(ca["extra_phonemes"])
        assert got == c["expected"]


def test_convert_to_textgrid_method(tmp_path):
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textgrid")
    d = json.load(open(os.path.join(here, "LJ001-0002.vs.json")))
    al = _host_aligner()
    out = tmp_path / "x.TextGrid"
    text = al.convert_to_textgrid(d, output_file=str(out))
    assert text == open(os.path.join(here, "LJ001-0002.TextGrid"), encoding="utf-8").read()
    assert out.read_text(encoding="utf-8") == text


def test_complete_target_coverage_matches_reference(host_gold):
    """ensure_target_coverage with ensure_completeness=True (core.py:462-679): 300 synthetic aligner outputs with
    missing / repeated / invalid targets and trailing silences, expected rows from the reference."""
    from bournemouth_forced_aligner_amd.coverage import ensure_target_coverage
    n_est = 0
    for i, c in enumerate(host_gold["complete"]):
        rows = [tuple(r) for r in c["rows"]]
        if c["expected"] == "raises":
            with pytest.raises(Exception):
                ensure_target_coverage([c["seq"]], [rows], seq_lens=[c["len"]], silence_class=c["sil"])
            continue
        got = ensure_target_coverage([torch.tensor(c["seq"])], [rows], seq_lens=[c["len"]], silence_class=c["sil"])[0]
        assert [list(r) for r in got] == c["expected"], f"case {i}"
        n_est += sum(1 for r in got if r[4])
    assert n_est > 500


def test_batch_shaping_equals_per_row_convert_to_ms():
    """_shape_rows (float32 ms arithmetic over the whole batch) against convert_to_ms row by row in the tensor mode
    the reference ends up in (utils.py:115-149 with a 0-dim tensor spectral length), incl. the arrays mode."""
    from types import SimpleNamespace
    from bournemouth_forced_aligner_amd.utils import convert_to_ms
    rng = np.random.default_rng(5)
    al = _host_aligner()
    B, cap = 9, 12
    cnt = rng.integers(0, cap + 1, B).astype(np.int32)
    segs = np.zeros((B, cap, 4), np.int32)
    for b in range(B):
        st = np.sort(rng.integers(0, 900, cnt[b]))
        segs[b, :cnt[b], 0] = rng.integers(0, 66, cnt[b])
        segs[b, :cnt[b], 1] = st
        segs[b, :cnt[b], 2] = st + rng.integers(1, 30, cnt[b])
        segs[b, :cnt[b], 3] = np.arange(cnt[b])
    conf = rng.random((B, cap)).astype(np.float32)
    spec = rng.integers(1, 1000, B).tolist()
    spec[3] = 0
    wav = (np.asarray(spec) * 268 + rng.integers(0, 200, B)).tolist()
    offs = [0.0, 0.5, 1.25, 3.0, 10.1, 0.333, 7.0, 2.5, 100.25]
    est = [[bool(x) for x in rng.integers(0, 2, cnt[b])] for b in range(B)]
    res = SimpleNamespace(segs=torch.from_numpy(segs.copy()), seg_count=torch.from_numpy(cnt.copy()))
    for offsets in (offs, 0.75):
        rows = al._shape_rows(res, torch.from_numpy(conf.copy()), est, spec, wav, offsets, False)
        arr = al._shape_rows(res, torch.from_numpy(conf.copy()), est, spec, wav, offsets, True)
        for b in range(B):
            off = offsets[b] if isinstance(offsets, list) else offsets
            six = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), est[b][i], float(conf[b, i])) for i, r in enumerate(segs[b, :cnt[b]])]
            exp = sorted(convert_to_ms(six, torch.tensor(spec[b]), off, wav[b], 16000), key=lambda x: x[6])
            assert rows[b] == exp, f"item {b}"
            n = int(arr["count"][b])
            got = list(zip(*(arr["rows"][b, :n, k].tolist() for k in range(4)), arr["is_estimated"][b, :n].tolist(),
                           arr["confidence"][b, :n].tolist(), arr["start_ms"][b, :n].tolist(), arr["end_ms"][b, :n].tolist()))
            assert got == exp


def _bench_json(argv, timeout=600):
    """Run bench.py as the driver does (a fresh process) and parse the one JSON line rank 0 prints."""
    import json
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_dry_run_spawns_two_ranks_that_agree_on_the_partition():
    """`bench.py --gpus 2` started by hand must itself become two ranks; --dry-run = gloo, no GPU work:
    the C4 plan (lengths -> LPT partition) is computed by both ranks and compared, and the packed gather runs."""
    out = _bench_json(["--gpus", "2", "--dry-run", "--config", "c4", "--global-batch", "4096"])
    assert out["dry_run"] and out["n_gpus"] == 2
    assert len(out["ranks"]) == 2 and len({r["pid"] for r in out["ranks"]}) == 2
    assert [r["rank"] for r in out["ranks"]] == [0, 1]
    assert out["shard_agree"] and out["gather_ok"]
    assert sum(out["shard_sizes"]) == 4096 and out["load_imbalance_max_over_mean"] < 1.01
    head = _bench_json(["--gpus", "2", "--dry-run"])
    assert head["n_gpus"] == 2 and head["gather_ok"]


def test_c4_generator_is_a_function_of_the_global_index_only():
    """tools/synth.py: an utterance's posteriors must not depend on the batch it is synthesised in (that is what lets
    each rank make only its shard and rank 0 re-make the parity sample)."""
    from tools import synth
    T, S = synth.c4_lengths(5000, seed=1004)
    assert T.min() >= 200 and T.max() <= 3000 and (S == np.maximum(1, T // 25)).all()
    T2, _ = synth.c4_lengths(100, seed=1004)
    np.testing.assert_array_equal(T[:100], T2)
    small = np.argsort(T)[:6]
    a_idx, b_idx = small[[0, 2, 4, 5]], small[[5, 1, 2]]
    la, ta = synth.c4_utterances(a_idx, T[a_idx], S[a_idx], 67, 1004, "cpu")
    lb, tb = synth.c4_utterances(b_idx, T[b_idx], S[b_idx], 67, 1004, "cpu", Tpad=400, Spad=20)
    for (i, j) in ((3, 0), (1, 2)):
        g = a_idx[i]
        assert g == b_idx[j]
        assert torch.equal(la[i, :T[g]], lb[j, :T[g]]) and torch.equal(ta[i, :S[g]], tb[j, :S[g]])
    ca, cb = synth.input_checksum(la, T[a_idx]), synth.input_checksum(lb, T[b_idx])
    assert int(ca[3]) == int(cb[0]) and int(ca[1]) == int(cb[2]) and int(ca[0]) != int(ca[1])
    # a planted path: every token id appears as the arg-max of some frame, in order
    am = la[0, :T[a_idx[0]]].argmax(-1).numpy()
    runs = am[np.r_[True, am[1:] != am[:-1]]]
    assert [int(x) for x in runs if x != 66] == [int(x) for x in ta[0, :S[a_idx[0]]]]


def test_headline_k1_keeps_eight_waves_per_simd():
    """The headline runs in k_dp4w<2> (and small / short batches in <1>, <3>): producer + consumer pairs fill the machine
    only at 8 waves per SIMD, i.e. <= 64 VGPRs per role with nothing spilled to scratch inside the roles; LDS must let 16
    workgroups share a CU (<= 10 KB).  The compiler's resource remark is the check (hipcc cross-compiles without a GPU)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bournemouth-forced-aligner_amd", "csrc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                          "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "-c",
                          os.path.join(csrc, "bfa_dp_nk5_p4.hip"), "-o", os.devnull],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {}
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?VGPRs Spill: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)",
                         out.stderr, re.S):
        res[m.group(1)] = tuple(int(m.group(k)) for k in (2, 3, 4, 5))
    for rw in (1, 2, 3):
        key = [k for k in res if f"k_dp4wILi{rw}ELi4ELi3ELb0E" in k]
        assert key, sorted(res)
        vgpr, vspill, occ, lds = res[key[0]]
        assert vgpr <= 64 and vspill == 0 and occ == 8 and lds <= 10240, (rw, res[key[0]])


def test_chop_wav_is_the_reference_single_segment_front_end(capsys):
    """core.py:288-328: the public chop_wav / _rms_normalize the reference's examples and tests call."""
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    al = PhonemeTimestampAligner(preset=None, device="cpu")
    g = torch.Generator().manual_seed(5)
    wav = torch.randn(2, 40000, generator=g)
    row, n, code = al.chop_wav(wav, 1000, 9000)
    assert code == 0 and n == 8000 and row.shape == (al.wav_len_max,)
    mono = wav[:, 1000:9000].mean(dim=0)
    want = mono / mono.square().mean().sqrt()
    assert torch.equal(row[:n], want) and float(row[n:].abs().max()) == 0.0
    assert torch.equal(al._rms_normalize(mono), want)
    assert torch.equal(al._rms_normalize(torch.zeros(16)), torch.zeros(16))
    # longer than the model's window: cut
    long = torch.randn(1, al.wav_len_max + 500, generator=g)
    row, n, code = al.chop_wav(long, 0, al.wav_len_max + 500)
    assert code == 0 and n == al.wav_len_max
    # too short a request (also end = -1), and a clip that ends early: the reference's codes and messages
    assert al.chop_wav(wav, 0, 100) == (None, None, -1)
    assert al.chop_wav(wav, 0, -1) == (None, None, -1)
    assert "ERROR: Segment too short: -1 frames, minimum required is %d frames." % al.seg_duration_min_samples in capsys.readouterr().out
    assert al.chop_wav(wav, 39990, 45000) == (None, None, -2)
    assert "Wav shape is too small: torch.Size([2, 10]), start_frame: 39990, end_frame: 45000" in capsys.readouterr().out
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refload
    if refload.available():   # live: the reference's own method on the same clips
        ref = refload.core_aligner()
        for lo, hi in ((1000, 9000), (0, 100), (0, -1), (39990, 45000), (0, 40000)):
            a, b = al.chop_wav(wav, lo, hi), ref.chop_wav(wav, lo, hi)
            assert a[1:] == b[1:]
            assert (a[0] is None and b[0] is None) or torch.equal(a[0], b[0])
        capsys.readouterr()


def test_bench_gpus_8_dry_run_c4_eight_ranks_agree_and_gather():
    """The first 8-GPU run must not be the first time the code sees 8 ranks: `bench.py --gpus 8 --config c4`
    becomes eight processes (gloo here), each pinned to the device of its LOCAL_RANK, that agree on the LPT partition of the
    fixed batch, and the packed gather puts every utterance's record where it belongs."""
    out = _bench_json(["--gpus", "8", "--dry-run", "--config", "c4", "--global-batch", "4096"], timeout=900)
    assert out["dry_run"] and out["n_gpus"] == 8
    rk = out["ranks"]
    assert [r["rank"] for r in rk] == list(range(8)) and len({r["pid"] for r in rk}) == 8
    assert [r["local_rank"] for r in rk] == list(range(8)) and [r["would_pin"] for r in rk] == [f"cuda:{i}" for i in range(8)]
    assert all(r["ipc_mode_legacy"] == "0" for r in rk)  # dmabuf IPC: RCCL needs it on this driver
    assert out["shard_agree"] and out["gather_ok"]
    assert sum(out["shard_sizes"]) == 4096 and min(out["shard_sizes"]) > 0 and out["load_imbalance_max_over_mean"] < 1.01
    # fewer utterances than ranks: empty shards take part in the gather
    tiny = _bench_json(["--gpus", "8", "--dry-run", "--config", "c4", "--global-batch", "5"], timeout=900)
    assert tiny["n_gpus"] == 8 and sorted(tiny["shard_sizes"]) == [0, 0, 0, 1, 1, 1, 1, 1] and tiny["gather_ok"]


def _gloo_worker8(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bournemouth_forced_aligner_amd.sharding import gather_results
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # uneven shards by hand: rank r owns r utterances (rank 0 -- the destination -- owns none) with capacity 3 + 2 r
    owner = np.concatenate([np.full(r, r) for r in range(world)])
    n = len(owner)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n)                      # global position of the k-th owned utterance
    capmax = 3 + 2 * (world - 1)
    segs = rng.integers(0, 900, size=(n, capmax, 4)).astype(np.int32)
    conf = rng.random((n, capmax)).astype(np.float32)
    mine = perm[owner == rank]
    cap = 3 + 2 * rank
    cnt = np.minimum(rng.integers(0, capmax + 1, size=n), 3 + 2 * owner[np.argsort(perm)]).astype(np.int32)  # count <= owner's capacity
    out = gather_results(torch.from_numpy(segs[mine][:, :cap]), torch.from_numpy(cnt[mine]), torch.from_numpy(conf[mine][:, :cap]),
                         torch.from_numpy(mine.astype(np.int64)), n, dst=0)
    if rank == 0:
        gs, gc, gf = out.to_padded(capmax)
        ok = gs.shape == (n, capmax, 4) and not out.overflowed()
        # the payload is the packed tuples, not n x capmax padded rows: 8 ranks x (header + tables + 20 B per tuple of the fullest bound)
        ok &= out.records.numel() * 4 <= 8 * (32 + 3 * 4 * 8 + 20 * 7 * capmax + 64)
        for i in range(n):
            c = int(cnt[i])
            ok &= int(gc[i]) == c and bool((gs[i, :c].numpy() == segs[i, :c]).all()) and bool((gf[i, :c].numpy() == conf[i, :c]).all())
            t, cf = out.rows(i)
            ok &= bool((t == segs[i, :c]).all()) and bool((cf == conf[i, :c]).all())
        open(os.path.join(tmp, "ok8"), "w").write("1" if ok else "0")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_8_gather_with_an_empty_shard_and_unequal_capacities(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert open(os.path.join(str(tmp_path), "ok8")).read() == "1"


def test_relaunch_under_torchrun_command_and_environment(monkeypatch):
    """`python bench.py --gpus N` re-executes itself under torch.distributed.run: one rank per GPU, rendezvous on 127.0.0.1,
    and the environment multi-process GPU work needs on this driver (dmabuf IPC); the library needs no queue setting."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert bench.relaunch_under_torchrun(8, ["--gpus", "8", "--config", "c4"]) == 0
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--config", "c4"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "GPU_MAX_HW_QUEUES" not in env


def test_lazy_row_lists_behave_like_the_reference_list_of_lists():
    """decode_alignments returns a LazyRowLists (forced_alignment.py:871,908 return list[list[tuple]]): indexing, negative
    indices, slices, iteration, len, equality with plain lists (both ways), mutation of a row, tolist()."""
    from bournemouth_forced_aligner_amd.forced_alignment import LazyRowLists, _ROW4, rows_as_tuple_lists
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 6, size=40)
    rec = np.zeros(int(counts.sum()), _ROW4)
    for k in ("phoneme", "start", "end", "target_idx"):
        rec[k] = rng.integers(0, 1000, size=rec.shape[0])
    plain = rows_as_tuple_lists(rec, counts)
    lazy = LazyRowLists(rec, counts)
    assert len(lazy) == 40 and lazy == plain and plain == lazy and not (lazy != plain)
    assert lazy[7] == plain[7] and lazy[-1] == plain[-1] and lazy[3:9] == plain[3:9] and lazy[::-7] == plain[::-7]
    assert [r for r in lazy] == plain and list(lazy) == plain and lazy.tolist() == plain
    assert all(isinstance(t, tuple) and all(isinstance(v, int) for v in t) for r in lazy for t in r)
    fresh = LazyRowLists(rec, counts)
    fresh[5].append((1, 2, 3, 4))          # a row is a real list: kept once it has been built
    assert fresh[5][-1] == (1, 2, 3, 4) and fresh.tolist()[5][-1] == (1, 2, 3, 4) and fresh != plain
    with pytest.raises(IndexError):
        lazy[40]
    assert LazyRowLists(rec[:0], []) == [] and len(LazyRowLists(rec[:0], [0, 0])) == 2


def test_hint_and_launcher_agree_on_the_launch_layout():
    """ViterbiDecoder.hint_and_path (the host-side hint and the layout it is written for) against bfa_call_path (the
    library's own decision from the same shapes and hint; host arithmetic, no device): per-class kernels, the one-kernel
    mixed-length path, the one-kernel path of small single-class calls.  The constants on both sides are the same ones."""
    import ctypes
    from bournemouth_forced_aligner_amd import _lib
    from bournemouth_forced_aligner_amd.forced_alignment import ViterbiDecoder
    L = _lib.lib()
    src = open(os.path.join(ROOT, "bournemouth-forced-aligner_amd", "csrc", "bfa_types.hpp")).read()
    assert f"ONE_MAX_BATCH = {_lib.ONE_MAX_BATCH};" in src and f"#define BFA_MIX_MIN_BATCH {_lib.MIX_MIN_BATCH}\n" in src
    assert _lib.ONE_HINT_MAX_BATCH <= _lib.ONE_MAX_BATCH
    rng = np.random.default_rng(17)
    seen = set()
    for trial in range(400):
        C = int(rng.choice([67, 17, 40]))
        vd = ViterbiDecoder(C - 1, 0, silence_anchors=10, truly_forced=True)
        B = int(rng.choice([1, 2, 3, 16, 63, 64, 65, 300, 1024, 1025, 4096]))
        kind = rng.integers(0, 5)
        if kind == 0:      # uniform lengths (the headline shape and smaller ones)
            T = np.full(B, int(rng.choice([200, 600, 1000, 2000])))
            S = np.full(B, int(rng.choice([5, 20, 40, 63, 64, 70])))
        elif kind == 1:    # mixed lengths, S = T // 25 (C4)
            T = rng.integers(200, 3001, size=B)
            S = np.maximum(1, T // 25)
        elif kind == 2:    # short sentences of one class
            T = rng.integers(300, 420, size=B)
            S = np.full(B, int(rng.choice([16, 20, 30])))
        elif kind == 3:    # dense targets (strides below 4, proportional, too short) and empty ones
            T = rng.integers(20, 400, size=B)
            S = np.minimum(rng.integers(0, 120, size=B), T + 3)
        else:              # long paths (wide classes)
            T = rng.integers(1500, 3000, size=B)
            S = rng.integers(100, 400, size=B)
        has_sil = bool(rng.integers(0, 4) == 0)
        simple = bool(rng.integers(0, 8) == 0)
        pad = int(rng.choice([0, 0, 0, 7, 40]))
        Smax = max(1, int(S.max()) + pad)
        mask, path = vd.hint_and_path(T, S, has_sil, simple=simple, n_classes=C, Smax=Smax)
        p = vd._params(True, True, not simple, simple)
        p.class_mask = int(mask)
        got = L.bfa_call_path(B, int(T.max()), Smax, C, ctypes.byref(p), 1)
        assert got == path, (trial, C, B, kind, has_sil, simple, pad, hex(mask), got, path)
        seen.add(got)
    assert seen == {_lib.PATH_CLASS_KERNELS, _lib.PATH_MIXED, _lib.PATH_ONE_KERNEL}
    # T_len == NULL: every utterance has Tmax frames -> never the mixed-length path, with or without a hint
    vd = ViterbiDecoder(66, 0, silence_anchors=10, truly_forced=True)
    p = vd._params(True, True, True)
    p.class_mask = _lib.HINT_NO_SILENCE_TARGETS
    assert L.bfa_call_path(4096, 1000, 40, 67, ctypes.byref(p), 1) == _lib.PATH_MIXED
    assert L.bfa_call_path(4096, 1000, 40, 67, ctypes.byref(p), 0) == _lib.PATH_CLASS_KERNELS
