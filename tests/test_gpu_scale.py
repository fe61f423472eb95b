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
