#!/usr/bin/env python3
"""bench.py -- aligned frames/s of the MI355X forced-alignment core on synthetic ph66 posteriors.

A "step" is one pass of the hot path (bfa_align_batch: planning + K1 banded Viterbi forward with
fused boost/log_softmax/floor + K2 backtrace + K3 run-length encoding) over one batch of
device-resident log-probabilities.  Workload at every N: BASELINE.json configs[2]
("batch=4096 T=1000 |tokens|=40 ph66") PER GPU (weak scaling; utterances are independent, so ranks
share nothing and the data path has no collective).

One JSON line is printed by rank 0.  `roofline` prices the dominant kernel (K1) with the
algorithmic bytes of SURVEY.md section 8(d): 4*C + ceil(L/4) + 8 bytes per frame.
`cpu_baseline` times the C restatement of the reference (oracle/, kind "port") on the host.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NO_WINDOW = False
HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def synth_batch(B, T, S, C, seed, device, peak=9.0):
    """Planted-path posteriors: tokens iid uniform on 1..C-2 (no SIL, no blank),
    random monotone segmentation with >= 2 frames per token, logits = N(0,1) + peak*onehot(planted),
    log_probs = log_softmax(logits).  Generated on the device with a seeded torch generator."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = C - 1
    toks = torch.randint(1, C - 1, (B, S), generator=g, device=device)
    extra = T - 2 * S
    assert extra >= 0
    cuts, _ = torch.sort(torch.randint(0, extra + 1, (B, 2 * S), generator=g, device=device), dim=1)
    zeros = torch.zeros((B, 1), dtype=cuts.dtype, device=device)
    full = torch.full((B, 1), extra, dtype=cuts.dtype, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, full], dim=1), dim=1)  # [B, 2S+1] gap,tok,gap,tok,...,gap
    sizes[:, 1::2] += 2
    ends = torch.cumsum(sizes, dim=1)  # slot k covers [ends[k-1], ends[k])
    t = torch.arange(T, device=device).unsqueeze(0).expand(B, T).contiguous()
    slot = torch.searchsorted(ends, t, right=True)  # [B,T] in 0..2S
    is_tok = (slot % 2) == 1
    tok_idx = torch.clamp((slot - 1) // 2, 0, S - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    logits = torch.randn((B, T, C), generator=g, device=device, dtype=torch.float32)
    logits.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, T, 1), peak, device=device))
    lp = torch.log_softmax(logits, dim=-1)
    return lp, toks.to(torch.int32)


def synth_ragged(B, Tlo, Thi, C, seed, device, peak=9.0):
    """BASELINE.json configs[3] shape: T ~ U{Tlo..Thi}, S = max(1, T // 25), padded to Thi / max S.  Every
    utterance gets its own planted path over its own T frames and S tokens (same construction as synth_batch)."""
    gc = torch.Generator(device="cpu")
    gc.manual_seed(seed)
    T_len = torch.randint(Tlo, Thi + 1, (B,), generator=gc)
    S_len = torch.clamp(T_len // 25, min=1)
    Tmax, Smax = int(T_len.max()), int(S_len.max())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = C - 1
    Td, Sd = T_len.to(device), S_len.to(device)
    toks = torch.randint(1, C - 1, (B, Smax), generator=g, device=device)
    extra = (Td - 2 * Sd).unsqueeze(1)  # [B,1] frames not forced to a token
    u = torch.rand((B, 2 * Smax), generator=g, device=device)
    cuts = torch.floor(u * (extra + 1).to(u.dtype)).to(torch.int64)
    k = torch.arange(2 * Smax, device=device).unsqueeze(0)
    cuts = torch.where(k < 2 * Sd.unsqueeze(1), cuts, extra.expand(-1, 2 * Smax))  # unused slots get no frames
    cuts, _ = torch.sort(cuts, dim=1)
    zeros = torch.zeros((B, 1), dtype=cuts.dtype, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, extra], dim=1), dim=1)  # [B, 2Smax+1] gap,tok,gap,tok,...,gap
    tokslot = torch.arange(Smax, device=device).unsqueeze(0) < Sd.unsqueeze(1)
    sizes[:, 1::2] += 2 * tokslot.to(sizes.dtype)
    ends = torch.cumsum(sizes, dim=1)
    t = torch.arange(Tmax, device=device).unsqueeze(0).expand(B, Tmax).contiguous()
    slot = torch.searchsorted(ends, t, right=True)
    is_tok = ((slot % 2) == 1) & (t < Td.unsqueeze(1))
    tok_idx = torch.clamp((slot - 1) // 2, 0, Smax - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    logits = torch.randn((B, Tmax, C), generator=g, device=device, dtype=torch.float32)
    logits.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, Tmax, 1), peak, device=device))
    lp = torch.log_softmax(logits, dim=-1)
    return lp, toks.to(torch.int32), T_len.to(torch.int32), S_len.to(torch.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--tokens", type=int, default=40)
    ap.add_argument("--classes", type=int, default=67)
    ap.add_argument("--cpu-sample", type=int, default=8192, help="utterances timed on the host oracle (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-window", action="store_true", help="A/B: full state layout instead of the sliding window")
    ap.add_argument("--win-frames", type=int, default=0, help="A/B: window frame limit (bfa_params.reserved[2]), 0 = default")
    ap.add_argument("--win-tokens", type=int, default=0, help="A/B: window token limit (bfa_params.reserved[1]), 0 = default")
    ap.add_argument("--ragged", action="store_true",
                    help="side measurement: mixed-length batch T~U{200..3000}, S=T//25 (BASELINE.json configs[3] per-GPU shard)")
    args = ap.parse_args()
    global NO_WINDOW
    NO_WINDOW = bool(args.no_window)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch, _lib

    B, T, S, C = args.batch, args.frames, args.tokens, args.classes
    blank, sil = C - 1, 0
    if args.ragged:
        return ragged_main(args, dev, rank, world)
    # two distinct batches so consecutive steps never stream the same 1.1 GB (> the 256 MB Infinity Cache anyway)
    bufs = [synth_batch(B, T, S, C, 1003 + 17 * rank + 1000 * i, dev) for i in range(2)]
    T_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    S_len = torch.full((B,), S, dtype=torch.int32, device=dev)
    au = AlignmentUtils(blank_id=blank, silence_id=sil)  # reference defaults: anchors 10, boost, floor, truly_forced
    lib = _lib.lib()
    h = _lib.handle(local_rank)

    # the lengths are known on the host (constant here), and the synthetic targets never contain SIL:
    # tell the library which K1 register class occurs so that it does not launch the empty ones
    hint = au.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=(None if NO_WINDOW else C))

    def step(i):
        lp, tk = bufs[i % 2]
        return au.decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint)

    for i in range(args.warmup):
        res = step(i)
    torch.cuda.synchronize()
    st = res.status.cpu().numpy()
    if not os.environ.get("BFA_HIP_LIBRARY"):  # (kernel-time experiments with a stubbed role produce garbage)
        assert (st == 0).all(), f"alignment failed on the bench workload: {np.unique(st)}"

    # K1 of every step is bracketed with HIP events on the launch stream (measured: no effect on the step time;
    # BFA_BENCH_K1_EVERY=n samples every n-th step instead)
    k1_every = int(os.environ.get("BFA_BENCH_K1_EVERY", "1"))
    lib.bfa_profile_enable(h, k1_every)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    lib.bfa_profile_enable(h, 0)
    k1 = (ctypes.c_float * max(1, args.steps))()
    nk1 = lib.bfa_profile_collect(h, k1, args.steps)
    k1_ms = float(np.mean([k1[i] for i in range(nk1)])) if nk1 > 0 else float("nan")

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # confidence pass (utils._calculate_confidences), timed separately: it is a separate reference call
    lp0, _ = bufs[0]
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(5):
        conf, _st = calculate_confidences_batch(lp0, res.segs, res.seg_count)
    torch.cuda.synchronize()
    conf_ms = (time.perf_counter() - c0) / 5 * 1e3

    # final gather of the (small) result records over RCCL, outside the timed steps
    gather_ms = None
    if dist is not None:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        out = [torch.empty_like(res.segs) for _ in range(world)] if rank == 0 else None
        dist.gather(res.segs, out, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    frames_per_step = B * T
    L = 4 * S + 1
    bytes_per_frame = 4 * C + (L + 3) // 4 + 8
    value = world * frames_per_step * args.steps / elapsed
    achieved = frames_per_step * bytes_per_frame / (k1_ms * 1e-3) / 1e9 if k1_ms == k1_ms else None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:  # timed at N=1 only
        from oracle import oracle as ora
        prm = ora.make_params(blank, sil)
        per = min(args.cpu_sample, B)
        nbuf = max(1, min(len(bufs), (args.cpu_sample + B - 1) // B))
        lp_w = bufs[0][0][:4].cpu().numpy()
        ora.decode_alignments(lp_w, bufs[0][1][:4].cpu().numpy(), [T] * 4, [S] * 4, prm, seg_cap=S + 2)  # warm
        w = 0.0
        n_total = 0
        mism = 0
        for bi in range(nbuf):
            lp_h = bufs[bi][0][:per].cpu().numpy()
            tk_h = bufs[bi][1][:per].cpu().numpy()
            w0 = time.perf_counter()
            exp = ora.decode_alignments(lp_h, tk_h, [T] * per, [S] * per, prm, seg_cap=S + 2)
            w += time.perf_counter() - w0
            n_total += per
            # the bench doubles as a full-size parity check on that sample
            got = step(bi)
            torch.cuda.synchronize()
            gs = got.segs[:per].cpu().numpy()
            gc = got.seg_count[:per].cpu().numpy()
            for b in range(per):
                if gc[b] != exp["seg_count"][b] or not (gs[b, :gc[b]] == exp["seg"][b, :gc[b]]).all():
                    mism += 1
        cpu = {"value": n_total * T / w, "unit": "aligned frames/s", "cores": 1, "kind": "port",
               "sample": f"{n_total} utterances of T={T} S={S} C={C} (the bench batches themselves), "
                         f"oracle/bfa_oracle.c single thread, {w:.1f} s",
               "parity_mismatching_utterances": mism}
        # the same restatement on all host cores (one utterance stream per thread; the ctypes call releases the GIL)
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncores = max(1, min(len(os.sched_getaffinity(0)), 256))
            lp_h = bufs[0][0][:per].cpu().numpy()
            tk_h = bufs[0][1][:per].cpu().numpy()
            chunk = (per + ncores - 1) // ncores

            def run(k):
                lo, hi = k * chunk, min(per, (k + 1) * chunk)
                if hi > lo:
                    ora.decode_alignments(lp_h[lo:hi], tk_h[lo:hi], [T] * (hi - lo), [S] * (hi - lo), prm, seg_cap=S + 2)
            with ThreadPoolExecutor(ncores) as ex:
                a0 = time.perf_counter()
                list(ex.map(run, range(ncores)))
                wa = time.perf_counter() - a0
            cpu["all_cores"] = {"value": per * T / wa, "cores": ncores,
                                "sample": f"{per} utterances over {ncores} threads, {wa:.2f} s"}
        except Exception as e:  # the single-thread figure above is the contract; this one is extra
            cpu["all_cores"] = {"error": repr(e)}

    # HBM bytes of one K1 launch from the PMC passes of tools/profile.sh (committed under profiles/), if it was
    # taken on this workload; FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_k1_traffic.json")
    if os.path.exists(tpath) and (B, T, S, C) == (4096, 1000, 40, 67):
        traffic = json.load(open(tpath))["traffic_bytes_per_launch"]

    if rank == 0:
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors", "value": value,
            "unit": "aligned frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch={B} T={T} |tokens|={S} ph66 (C={C}) per GPU, reference-default flags "
                                   f"(boost+floor+truly_forced, anchors=10, no SIL in targets -> standard mode)",
                       "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per K1 launch (rocprofv3 PMC, profiles/r01_k1_traffic.json)",
                         "algorithmic_bytes_per_launch": frames_per_step * bytes_per_frame,
                         "kernel": "k_dp4w<2,4,3> (K1 banded Viterbi forward, sliding-window consumer; HIP events also span the k_dp4_redo launch)", "kernel_ms": k1_ms, "kernel_ms_samples": int(nk1),
                         "algorithmic_bytes_per_frame": bytes_per_frame},
            "cpu_baseline": cpu,
            "confidence_pass_ms": conf_ms,
            "gather_ms": gather_ms,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def ragged_main(args, dev, rank, world):
    """Not the headline line: throughput on a mixed-length shard (prints its own JSON; its parity check against the
    oracle is tests/test_gpu_parity.py::test_mixed_length_shard_parity)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    C, B = args.classes, args.batch
    lp, tk, T_len, S_len = synth_ragged(B, 200, 3000, C, 1004 + rank, dev)
    au = AlignmentUtils(blank_id=C - 1, silence_id=0)
    au.viterbi_decoder.window_max_frames = args.win_frames or None
    au.viterbi_decoder.window_max_tokens = args.win_tokens or None
    hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=(None if NO_WINDOW else C))
    Td, Sd = T_len.to(dev), S_len.to(dev)
    for _ in range(2):
        res = au.decode_alignments_device(lp, tk, Td, Sd, class_mask=hint)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = au.decode_alignments_device(lp, tk, Td, Sd, class_mask=hint)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    frames = int(T_len.sum())
    if rank == 0:
        print(json.dumps({"workload": f"ragged batch={B} T~U[200,3000] S=T//25 C={C}", "frames": frames,
                          "ms_per_step": el * 1e3, "frames_per_s": frames / el,
                          "status_ok": bool((res.status.cpu() == 0).all())}))


if __name__ == "__main__":
    main()
