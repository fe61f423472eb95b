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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun_bench_json(n, argv, timeout=1800):
    """what the driver does for N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + argv
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _c4_ranks_share_one_gpu(n, global_batch):
    out = _torchrun_bench_json(n, ["--config", "c4", "--oversubscribe", "--global-batch", str(global_batch), "--steps", "2",
                                   "--warmup", "1", "--parity-sample", str(global_batch)])
    assert out["n_gpus"] == n and len(out["ranks"]) == n
    assert all(r["device"] == "cuda:0" and r.get("oversubscribed") for r in out["ranks"])
    assert len({r["pid"] for r in out["ranks"]}) == n and sorted(r["rank"] for r in out["ranks"]) == list(range(n))
    assert out["group"] == {"backend": "gloo", "ranks_seen": n, "rccl_ranks_seen": None, "oversubscribed": True}
    assert len(out["shard_sizes"]) == n and sum(out["shard_sizes"]) == global_batch and min(out["shard_sizes"]) > 0
    assert out["shard_agree"] and len(out["rank_ms_per_step"]) == n and min(out["rank_ms_per_step"]) > 0
    ps = out["parity_sample"]   # EVERY utterance: re-synthesised on rank 0, its gathered record against the oracle
    assert ps["utterances"] == global_batch and ps["mismatching_utterances"] == 0 and ps["regenerated_inputs_differing"] == 0
    assert out["gather"]["payload_bytes"] > 0 and out["frames_per_step"] > global_batch * 200
    return out


def test_c4_two_ranks_share_one_gpu(gpu_device):
    """N = 2 with REAL GPU work: two processes on the one GPU of the box, a gloo group, every rank
    synthesises and aligns its LPT shard, packs its records on the device, ONE gather (records staged through pinned host
    memory: RCCL refuses two ranks on a device), rank 0 indexes the records and checks every utterance against the oracle."""
    _c4_ranks_share_one_gpu(2, 1024)


def test_c4_eight_ranks_share_one_gpu(gpu_device):
    """the same with the EIGHT ranks of a node: partition -> align -> pack -> gather -> index under N = 8, real results"""
    _c4_ranks_share_one_gpu(8, 2048)


def test_headline_two_ranks_share_one_gpu(gpu_device):
    """the weak-scaling mode's N > 1 leg with real alignments on both ranks: barriers, max over ranks, the gather of both
    ranks' records and the comparison of rank 0's gathered rows with its own results"""
    out = _torchrun_bench_json(2, ["--oversubscribe", "--batch", "512", "--steps", "4", "--warmup", "2", "--no-cpu",
                                   "--settle-ms", "0", "--min-timed-steps", "8"])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024 and out["gather_ms"] is not None
    assert out["group"]["backend"] == "gloo" and out["group"]["ranks_seen"] == 2 and out["value"] > 0


def test_c4_mode_world_size_1_nccl(gpu_device):
    """BASELINE.json configs[3] end to end at reduced global batch: plan -> LPT shard -> per-rank synthesis and
    alignment in length-sorted calls -> sharding.gather_results over an RCCL process group -> rank 0 re-synthesises a
    stratified sample (incl. the longest utterances) and compares the GATHERED records with the oracle."""
    # EVERY utterance of the reduced batch is compared, once as length-sorted calls of 512 (narrow length ranges: the per-class
    # kernels) and once as ONE call (the mixed-length path, k_mix)
    for chunk in ("512", "32768"):
        out = _bench_json(["--config", "c4", "--global-batch", "1536", "--chunk", chunk, "--steps", "2", "--warmup", "1",
                           "--parity-sample", "1536"])
        assert out["n_gpus"] == 1 and len(out["ranks"]) == 1 and "cuda:0" == out["ranks"][0]["device"]
        ps = out["parity_sample"]
        assert ps["utterances"] == 1536 and ps["mismatching_utterances"] == 0 and ps["regenerated_inputs_differing"] == 0
        assert ps["longest_T"] >= 2990  # the sample holds the longest utterances of the batch
        assert out["shard_sizes"] == [1536] and out["gather_ms"] is not None and out["value"] > 0
        assert out["scaling"] == "strong" and out["frames_per_step"] > 1536 * 200
        # the communicator itself saw one rank (all_reduce of ones over RCCL): what the driver's SCALE run prints per N
        assert out["rccl_ranks_seen"] == 1 and out["group"]["backend"] == "nccl"


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


def test_config_c3_full_batch_against_oracle(ora, gpu_device):
    """BASELINE.json configs[2] (the headline batch=4096, T=1000, |tokens|=40) aligned at full batch size, EVERY utterance
    of it compared with the oracle (framewise states, tuples, confidence bit patterns)."""
    _batch_vs_oracle(ora, gpu_device, 4096, 1000, 40, 1003, 4096)


def test_headline_mode_multi_rank_path_on_nccl(gpu_device):
    """The N > 1 leg of the headline mode (barriers, max over ranks, sharding.gather_results over RCCL, per-rank device
    list) with a one-rank `nccl` group: what the driver's `torch.distributed.run ... bench.py --gpus N` executes."""
    out = _bench_json(["--force-group", "--steps", "5", "--warmup", "2", "--no-cpu", "--settle-ms", "10",
                       "--min-timed-steps", "10"])
    assert out["n_gpus"] == 1 and out["gather_ms"] is not None and out["timing"]["windows"] == 2
    assert out["ranks"][0]["device"] == "cuda:0" and out["value"] > 1e9 and out["roofline"]["frac"] > 0.2
    assert out["timing"]["first_window"]["ms_per_step"] > 0


def test_bench_protocol_of_the_default_run(gpu_device):
    """`python bench.py --steps 20 --warmup 5` (what the driver runs): five windows of exactly 20 steps (>= 100 timed
    steps) with four batches in flight; the warm-up that really ran is reported; `roofline` is priced by
    the one-batch-in-flight leg of the same run (no launch overlaps another there), while the brackets of the reported
    windows -- which do overlap -- are kept under `in_flight`."""
    out = _bench_json(["--steps", "20", "--warmup", "5", "--no-cpu", "--settle-ms", "60"])
    r, t = out["roofline"], out["timing"]
    assert out["config"]["batches_in_flight"] == 4 and out["steps"] == 20 and out["warmup"] == 5
    assert t["windows"] == 5 and t["timed_steps_total"] == 100 and len(t["window_ms_per_step"]["all"]) == 5
    assert t["warmup_effective_steps"] >= 5 + 20 and t["warmup_requested_steps"] == 5
    assert abs(out["ms_per_step"] - np.mean(t["window_ms_per_step"]["all"])) < 1e-6 * out["ms_per_step"] + 1e-9
    assert r["kernel_ms_samples"] >= 20 and r["kernel_ms"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-9 and r["frac"] > 0.3 and out["value"] > 5e9
    fl = r["in_flight"]
    assert fl["launches_running_on_average"] > 1.2 and fl["busy_ms_per_launch"] < fl["launch_ms_stats"]["mean"]
    assert fl["busy_ms_per_launch"] <= out["ms_per_step"] * 1.02  # the kernel cannot be busy longer than the region
    # a lone kernel is not slower than a step of its own leg, and the leg's steps are plain stream-ordered calls
    assert r["kernel_ms"] < r["kernel_leg_ms_per_step"]
    # a step is alignment + confidence pass by default; the alignment-only figure of earlier rounds rides along
    ao = out["alignment_only"]
    assert out["step"].startswith("K1 + K2 + K3") and 0 < ao["ms_per_step"] < out["ms_per_step"] and ao["value"] > out["value"]
    assert r["confidence_pass_algorithmic_bytes"] > 64 * 4096 * 80 and 0.2 < r["whole_step_frac"] < 1.0
    print("4 in flight:", out["ms_per_step"], "ms/step; kernel alone", r["kernel_ms"], "frac", r["frac"], "whole step",
          r["whole_step_frac"], "busy/launch", fl["busy_ms_per_launch"])


def test_bench_realtext_mode(gpu_device):
    """`bench.py --config realtext` at reduced batch: both heads from raw logits with SIL in the targets through
    bfa_align_heads + bfa_postprocess + bfa_confidences, every sampled utterance against the oracle's whole chain."""
    out = _bench_json(["--config", "realtext", "--batch", "256", "--steps", "3", "--warmup", "1", "--settle-ms", "0",
                       "--min-timed-steps", "3", "--parity-sample", "64"])
    p = out["parity"]
    assert p["utterances"] == 64 and p["mismatching_utterances"] == {"ph66": 0, "groups": 0}
    assert p["confidence_beyond_1e-4"] == {"ph66": 0, "groups": 0} and p["tuples_compared"] > 64 * 2 * 30
    seg = out["segmented_utterances"]
    assert seg["ph66"] > 0.8 * seg["of"] and seg["groups"] > 0.8 * seg["of"]
    assert out["value"] > 0 and out["roofline"]["algorithmic_bytes_per_frame"] == 4 * 67 + 4 * 17 + 4 * 67 + 2 * 41 + 16


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
    # ... and with the confidence pass enqueued by the helper itself on the slot's stream (what a bench step is)
    for n in range(5):
        lp, tk, Tl, Sl = lens(work[n % len(work)])
        r = bif.submit(lp, tk, Tl, Sl, confidences=True)
        got.append((n % len(work), r, r.conf))
    bif.synchronize()
    torch.cuda.synchronize()
    for w, r, cf in got:
        cnt, segs, fph, rcf = ref[w]
        keep = np.arange(segs.shape[1])[None, :] < cnt[:, None]
        assert np.array_equal(r.seg_count.cpu().numpy(), cnt)
        assert np.array_equal(np.where(keep[:, :, None], r.segs.cpu().numpy(), 0), np.where(keep[:, :, None], segs, 0))
        assert np.array_equal(r.frame_phonemes.cpu().numpy(), fph)
        assert np.array_equal(np.where(keep, cf.cpu().numpy().view(np.int32), 0), np.where(keep, rcf.view(np.int32), 0))


def test_bench_c5proxy_mode(gpu_device):
    """`bench.py --config c5proxy --peak 5` at reduced batch: mixed-length segments up to 30 s, both heads from raw logits, SIL in
    the targets, posteriors soft enough that the silence anchoring mostly fails and long utterances end at the sentinel; the
    sampled utterances (incl. the longest) against the oracle's whole chain."""
    out = _bench_json(["--config", "c5proxy", "--batch", "96", "--steps", "2", "--warmup", "1", "--min-timed-steps", "2",
                       "--peak", "5", "--parity-sample", "48"])
    p = out["parity"]
    assert p["utterances"] >= 24 and p["mismatching_utterances"] == 0 and p["confidence_beyond_1e-4"] == 0
    assert out["value"] > 0 and out["softness"]["peak"] == 5.0 and out["frames_per_step"] > 96 * 300
    out = _bench_json(["--steps", "3", "--warmup", "1", "--no-cpu", "--settle-ms", "0", "--min-timed-steps", "3", "--batch", "2048",
                       "--peak", "6"])
    s = out["softness"]
    assert s["sample_share_at_sentinel"] > 0.9 and s["last_call"]["exact_done"] == 2048, s
