"""-m gpu: the multi-GPU driver paths of bench.py on the one GPU of the test box (world size 1, backend nccl = RCCL),
and BASELINE.json configs[1] / configs[2] at batch size against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench_json(argv, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_c4_mode_world_size_1_nccl(gpu_device):
    """BASELINE.json configs[3] end to end at reduced global batch: plan -> LPT shard -> per-rank synthesis and
    alignment in length-sorted calls -> sharding.gather_results over an RCCL process group -> rank 0 re-synthesises a
    stratified sample (incl. the longest utterances) and compares the GATHERED records with the oracle."""
    out = _bench_json(["--config", "c4", "--global-batch", "1536", "--chunk", "512", "--steps", "2", "--warmup", "1",
                       "--parity-sample", "96"])
    assert out["n_gpus"] == 1 and len(out["ranks"]) == 1 and "cuda:0" == out["ranks"][0]["device"]
    ps = out["parity_sample"]
    assert ps["utterances"] >= 90 and ps["mismatching_utterances"] == 0 and ps["regenerated_inputs_differing"] == 0
    assert out["shard_sizes"] == [1536] and out["gather_ms"] is not None and out["value"] > 0
    assert out["scaling"] == "strong" and out["frames_per_step"] > 1536 * 200


def _batch_vs_oracle(ora, dev, B, T, S, seed, n_check):
    sys.path.insert(0, ROOT)
    from tools.synth import synth_batch
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch
    C = 67
    lp, tk = synth_batch(B, T, S, C, seed, dev)
    au = AlignmentUtils(blank_id=C - 1, silence_id=0)
    res = au.decode_alignments_device(lp, tk, [T] * B, [S] * B)
    conf, _ = calculate_confidences_batch(lp, res.segs, res.seg_count)
    torch.cuda.synchronize()
    assert (res.status.cpu().numpy() == 0).all()
    pick = np.unique(np.linspace(0, B - 1, n_check).astype(np.int64))
    lp_h, tk_h = lp[pick].cpu().numpy(), tk[pick].cpu().numpy()
    exp = ora.decode_alignments(lp_h, tk_h, [T] * len(pick), [S] * len(pick), ora.make_params(C - 1, 0), seg_cap=S + 2)
    gs, gc = res.segs[pick].cpu().numpy(), res.seg_count[pick].cpu().numpy()
    gf, gi = res.frame_phonemes[pick].cpu().numpy(), res.frame_phonemes_idx[pick].cpu().numpy()
    cf = conf[pick].cpu().numpy()
    for k in range(len(pick)):
        c = int(exp["seg_count"][k])
        assert gc[k] == c and (gs[k, :c] == exp["seg"][k, :c]).all(), f"utterance {pick[k]}"
        assert (gf[k] == exp["frame_ph"][k]).all() and (gi[k] == exp["frame_idx"][k]).all()
        _, oc, _, _ = ora.confidences(lp_h[k], [tuple(r) for r in exp["seg"][k, :c]])
        assert np.array_equal(cf[k, :c].view(np.int32), oc.view(np.int32)), f"confidence bits, utterance {pick[k]}"


def test_config_c2_full_batch_against_oracle(ora, gpu_device):
    """BASELINE.json configs[1]: batch=256, T=600, |tokens|=20 -- every utterance of the batch."""
    _batch_vs_oracle(ora, gpu_device, 256, 600, 20, 1002, 256)


def test_config_c3_slice_against_oracle(ora, gpu_device):
    """BASELINE.json configs[2] (the headline batch=4096, T=1000, |tokens|=40) aligned at full batch size, a
    512-utterance slice of it compared with the oracle (framewise states, tuples, confidence bit patterns)."""
    _batch_vs_oracle(ora, gpu_device, 4096, 1000, 40, 1003, 512)


def test_headline_mode_multi_rank_path_on_nccl(gpu_device):
    """The N > 1 leg of the headline mode (barriers, max over ranks, sharding.gather_results over RCCL, per-rank device
    list) with a one-rank `nccl` group: what the driver's `torch.distributed.run ... bench.py --gpus N` executes."""
    out = _bench_json(["--force-group", "--steps", "5", "--warmup", "2", "--no-cpu", "--settle-ms", "10"])
    assert out["n_gpus"] == 1 and out["gather_ms"] is not None and len(out["rank_ms_per_step"]) == 1
    assert out["ranks"][0]["device"] == "cuda:0" and out["value"] > 1e9 and out["roofline"]["frac"] > 0.2
    assert out["settle"]["first_window"]["ms_per_step"] > 0


def test_tail_stream_pipeline_is_bitwise_identical(gpu_device):
    """include/bfa.h bfa_set_tail_stream: two decoders (own handle / workspace / outputs) taking turns on one stream with
    the walk and the run-length encoding of every call on a shared tail stream.  Every pipelined call must return exactly
    what the stream-ordered call returns -- standard mode, silence-anchored mode and a mixed-length batch -- and a
    confidence pass enqueued right behind a pipelined call must see its finished tuples."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from tools.synth import synth_batch, synth_ragged
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch
    dev = gpu_device
    work = []
    for k in range(3):  # standard mode, headline lengths
        lp, tk = synth_batch(384, 1000, 40, 67, 77 + k, dev)
        work.append((lp, tk, torch.full((384,), 1000, dtype=torch.int32), torch.full((384,), 40, dtype=torch.int32), 10))
    lp, tk, Tl, Sl = synth_ragged(256, 200, 1800, 67, 5, dev)  # several K1 classes, window reruns
    work.append((lp, tk, Tl, Sl, 10))
    rng = np.random.default_rng(3)  # silence-anchored mode
    lps, toks = [], []
    for _ in range(96):
        a, b, _ = cases.planted_case(rng, 700, 30, C=67, blank=66, peak=9.0, sil_rate=1 / 10, sil_len=(12, 40))
        lps.append(a); toks.append(b)
    a, b, Tl2, Sl2 = cases.pad_batch(lps, toks, 67, 66)
    work.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(torch.int32).to(dev),
                 torch.from_numpy(np.asarray(Tl2, np.int32)), torch.from_numpy(np.asarray(Sl2, np.int32)), 10))

    def fields(r):
        cnt = r.seg_count.cpu().numpy()
        segs = r.segs.cpu().numpy()
        keep = np.arange(segs.shape[1])[None, :] < cnt[:, None]
        return (cnt, np.where(keep[:, :, None], segs, 0), r.status.cpu().numpy(), r.mode.cpu().numpy(),
                r.frame_phonemes.cpu().numpy(), r.frame_phonemes_idx.cpu().numpy())

    plain = AlignmentUtils(blank_id=66, silence_id=0)
    ref = []
    for lp, tk, Tl, Sl, _ in work:
        r = plain.decode_alignments_device(lp, tk, Tl.to(dev), Sl.to(dev))
        cf, _ = calculate_confidences_batch(lp, r.segs, r.seg_count)
        torch.cuda.synchronize()
        ref.append(fields(r) + (cf.cpu().numpy(),))

    pipe = [AlignmentUtils(blank_id=66, silence_id=0) for _ in range(2)]
    for k, x in enumerate(pipe):
        x.viterbi_decoder.handle_slot = 4 + k
    tail = torch.cuda.Stream(device=dev)
    order = [0, 1, 2, 3, 4, 3, 0, 4, 2, 1, 4, 4, 3, 3]
    got = []
    for n, w in enumerate(order):  # nothing synchronises inside this loop
        lp, tk, Tl, Sl, _ = work[w]
        r = pipe[n % 2].decode_alignments_device(lp, tk, Tl.to(dev), Sl.to(dev), tail_stream=tail)
        got.append((w, r, None))
    torch.cuda.synchronize()
    for w, r, _ in got:
        for x, y in zip(fields(r), ref[w][:6]):
            assert np.array_equal(x, y), f"pipelined call on work item {w} differs from the stream-ordered call"
    # a consumer on the caller's stream right behind a pipelined call (the library orders it behind the pending tail)
    for n, w in enumerate([0, 4, 3]):
        lp, tk, Tl, Sl, _ = work[w]
        au = pipe[n % 2]
        r = au.decode_alignments_device(lp, tk, Tl.to(dev), Sl.to(dev), tail_stream=tail)
        cf, _ = calculate_confidences_batch(lp, r.segs, r.seg_count, handle_slot=au.viterbi_decoder.handle_slot)
        torch.cuda.synchronize()
        cnt = ref[w][0]
        keep = np.arange(cf.shape[1])[None, :] < cnt[:, None]
        assert np.array_equal(np.where(keep, cf.cpu().numpy().view(np.int32), 0), np.where(keep, ref[w][6].view(np.int32), 0))


def test_bench_default_keeps_three_batches_in_flight(gpu_device):
    """`python bench.py` (what the driver runs): three batches in flight on three streams (decoder, library handle and
    workspace each); the K1 launches overlap, the roofline prices the kernel by its busy time (union of the launch
    intervals).  With one batch in flight the busy time per launch is the launch duration."""
    out = _bench_json(["--steps", "20", "--warmup", "5", "--no-cpu", "--settle-ms", "60"])
    r = out["roofline"]
    assert out["config"]["batches_in_flight"] == 3 and r["kernel_ms_samples"] == 20
    assert r["launches_running_on_average"] > 1.2 and r["kernel_busy_ms_per_launch"] < r["kernel_ms"]
    assert r["kernel_busy_ms_per_launch"] <= out["ms_per_step"] * 1.02  # the kernel cannot be busy longer than the region
    assert r["frac"] > 0.3 and r["frac_per_launch_duration"] < r["frac"] and out["value"] > 5e9
    plain = _bench_json(["--steps", "20", "--warmup", "5", "--no-cpu", "--settle-ms", "60", "--inflight", "1"])
    q = plain["roofline"]
    assert plain["config"]["batches_in_flight"] == 1 and abs(q["kernel_busy_ms_per_launch"] - q["kernel_ms"]) < 0.02 * q["kernel_ms"]
    assert abs(q["frac"] - q["frac_per_launch_duration"]) < 0.02
    print("in flight 3:", out["ms_per_step"], r["kernel_busy_ms_per_launch"], r["kernel_ms"], "| 1:", plain["ms_per_step"], q["kernel_ms"])


def test_batches_in_flight_helper_matches_plain_calls(gpu_device):
    """bournemouth_forced_aligner_amd.BatchesInFlight (what bench.py times): three decoders / streams taking turns; every
    result must equal the plain call's, and `result.wait()` must order a consumer on the caller's stream behind it."""
    sys.path.insert(0, ROOT)
    from tools.synth import synth_batch, synth_ragged
    from bournemouth_forced_aligner_amd import AlignmentUtils, BatchesInFlight, calculate_confidences_batch
    dev = gpu_device
    work = [synth_batch(512, 1000, 40, 67, 300 + k, dev) + (None, None) for k in range(4)]
    lp, tk, Tl, Sl = synth_ragged(192, 200, 2600, 67, 9, dev)
    work.append((lp, tk, Tl.to(dev), Sl.to(dev)))
    plain = AlignmentUtils(blank_id=66, silence_id=0)

    def lens(w):
        lp, tk, Tl, Sl = w
        if Tl is None:
            Tl = torch.full((lp.shape[0],), lp.shape[1], dtype=torch.int32, device=dev)
            Sl = torch.full((lp.shape[0],), tk.shape[1], dtype=torch.int32, device=dev)
        return lp, tk, Tl, Sl

    ref = []
    for w in work:
        lp, tk, Tl, Sl = lens(w)
        r = plain.decode_alignments_device(lp, tk, Tl, Sl)
        cf, _ = calculate_confidences_batch(lp, r.segs, r.seg_count)
        torch.cuda.synchronize()
        ref.append((r.seg_count.cpu().numpy(), r.segs.cpu().numpy(), r.frame_phonemes.cpu().numpy(), cf.cpu().numpy()))
    bif = BatchesInFlight(66, 0, n=3, device=dev, first_handle_slot=8)
    got = []
    for n in range(11):
        lp, tk, Tl, Sl = lens(work[n % len(work)])
        r = bif.submit(lp, tk, Tl, Sl)
        r.wait()  # the confidence pass below runs on the caller's stream
        cf, _ = calculate_confidences_batch(lp, r.segs, r.seg_count)
        got.append((n % len(work), r, cf))
    bif.synchronize()
    torch.cuda.synchronize()
    for w, r, cf in got:
        cnt, segs, fph, rcf = ref[w]
        keep = np.arange(segs.shape[1])[None, :] < cnt[:, None]
        assert np.array_equal(r.seg_count.cpu().numpy(), cnt)
        assert np.array_equal(np.where(keep[:, :, None], r.segs.cpu().numpy(), 0), np.where(keep[:, :, None], segs, 0))
        assert np.array_equal(r.frame_phonemes.cpu().numpy(), fph)
        assert np.array_equal(np.where(keep, cf.cpu().numpy().view(np.int32), 0), np.where(keep, rcf.view(np.int32), 0))
