#!/usr/bin/env python3
"""bench.py -- aligned frames/s of the MI355X forced-alignment core on synthetic ph66 posteriors.

A "step" is one pass of the hot path (bfa_align_batch: planning + K1 banded Viterbi forward with
fused boost/log_softmax/floor + K2 backtrace + run-length encoding) over one batch of device-resident
log-probabilities.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      headline: BASELINE.json configs[2] ("batch=4096 T=1000 |tokens|=40 ph66") PER GPU, weak scaling.  The K timed
      steps are issued with three batches in flight (step i on stream i % 3 with its own decoder / library handle /
      workspace / outputs; --inflight 1 = one batch at a time): the latency-bound tail of a step runs beside the forward
      kernel of the next ones.  Every step does all of its work inside the timed region; value = frames / wall time.
  python bench.py --config c4 [--gpus N]
      BASELINE.json configs[3]: global B = 32768, T in [200,3000], S = T // 25 (seed 1004), LPT-sharded over the N
      ranks (sharding.shard_utterances), every rank synthesises and aligns only its shard, the result records are
      gathered to rank 0 (sharding.gather_results, RCCL) and a stratified sample is checked against the oracle there.

Ranks: one process per GPU.  Under `torch.distributed.run` (what the driver does for N > 1) the ranks come from the
environment; started directly with --gpus N > 1 the script re-executes itself under torch.distributed.run with N
processes.  `n_gpus` in the JSON line is the world size the process group reports, `ranks` lists the device of each.
`--dry-run` does the same launch with the gloo backend and no GPU work (partition + gather plumbing only; CPU test).

One JSON line is printed by rank 0.  `roofline` prices the dominant kernel (K1) with the algorithmic bytes of
SURVEY.md section 8(d): 4*C + ceil(L/4) + 8 bytes per frame, against the time K1 was running in the timed region
(union of the K1 launch intervals / launches; = the mean launch duration when one batch is in flight).  `cpu_baseline` times the C restatement of the
reference (oracle/, kind "port") on the host (N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.synth import (c4_lengths, c4_utterances, input_checksum, synth_batch,  # noqa: E402,F401
                         synth_ragged)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


# ------------------------------------------------------------------------------------------------ launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` started by hand: become N ranks (one per GPU) under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


class Ranks:
    """The process group of this run (or a single process without one)."""

    def __init__(self, args, need_group):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.dry = bool(args.dry_run)
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU implementation of the path)"
            assert self.local_rank < torch.cuda.device_count(), \
                f"rank {self.rank}: local rank {self.local_rank} but only {torch.cuda.device_count()} GPUs are visible"
            self.dev = torch.device("cuda", self.local_rank)
            torch.cuda.set_device(self.dev)
        if self.world > 1 or need_group:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()) if self.world == 1 else "29500")
            if self.dry:
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dist = dist
            self.world = dist.get_world_size()  # what the backend reports
        if args.gpus != self.world and self.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the process group has {self.world} ranks; reporting {self.world}",
                  file=sys.stderr)

    def describe(self):
        """[{rank, device, name, ...}] for every rank (gathered), so the line shows which GPUs really took part."""
        if self.dry:
            me = {"rank": self.rank, "device": "cpu (dry run)", "pid": os.getpid()}
        else:
            pr = torch.cuda.get_device_properties(self.dev)
            me = {"rank": self.rank, "device": f"cuda:{self.local_rank}", "name": pr.name,
                  "gcn_arch": getattr(pr, "gcnArchName", ""), "cus": pr.multi_processor_count,
                  "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")),
                  "hbm_gb": round(pr.total_memory / 2 ** 30, 1), "pid": os.getpid()}
        if self.dist is None:
            return [me]
        out = [None] * self.world
        self.dist.all_gather_object(out, me)
        return out

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x), [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device=self.dev)
        allt = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(allt, t)
        vals = [float(v.item()) for v in allt]
        return max(vals), vals

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def _stats(xs):
    xs = np.asarray(xs, np.float64)
    if xs.size == 0:
        return None
    return {"mean": float(xs.mean()), "min": float(xs.min()), "median": float(np.median(xs)),
            "p90": float(np.percentile(xs, 90)), "max": float(xs.max()), "n": int(xs.size)}


def _reference_cpu_record():
    """The real reference (forced_alignment.py) timed in the BUILD container by tools/time_reference.py; it cannot run
    on the GPU node (the reference never ships there), so its committed record is echoed beside the same-node port."""
    p = os.path.join(ROOT, "profiles", "reference_cpu_baseline.json")
    if not os.path.exists(p):
        return None
    try:
        rec = json.load(open(p))
        rec["where"] = "build container, NOT this node (tools/time_reference.py)"
        return rec
    except Exception:
        return None


# --------------------------------------------------------------------------------------------- headline
def headline_main(args, rk):
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch, _lib
    from bournemouth_forced_aligner_amd.sharding import gather_results
    dev, rank, world, dist = rk.dev, rk.rank, rk.world, rk.dist
    B, T, S, C = args.batch, args.frames, args.tokens, args.classes
    blank, sil = C - 1, 0
    # --pipeline n >= 2 (default 2): n decoders (own library handle, workspace, outputs) take turns on ONE stream and share
    # a tail stream (include/bfa.h, bfa_set_tail_stream): planning + K1 of step i+1 start as soon as K1 of step i has
    # ended, while the walk / run-length encoding of step i runs beside them.  --inflight n is the older form (whole steps
    # on n streams: their K1 kernels overlap each other and a launch's duration stops being a measure of the kernel).
    # Batches in flight (default 3): step i runs on stream i % 3 with decoder i % 3 (own library handle, workspace and
    # outputs), so the tail of a step (rerun launch, walk, run-length encoding: ~65 us of latency chains that leave the
    # machine mostly idle) and the ramp-down of its K1 overlap the K1 of the next steps.  `value` stays frames / wall time.
    # The K1 launches of different steps then overlap each other: a launch's own duration (kernel_ms) is no longer the
    # time the kernel needs for a batch, so the roofline prices the kernel by the time it was running AT ALL -- the union
    # of the launch intervals (bfa_profile_collect_spans) -- see `roofline` below.
    inflight = args.inflight if args.inflight is not None else 3
    npipe = args.pipeline if (inflight <= 1 and args.pipeline > 1) else 1
    nbuf = max(2, inflight, npipe)
    # distinct batches so consecutive steps never stream the same 1.1 GB (> the 256 MB Infinity Cache anyway)
    bufs = [synth_batch(B, T, S, C, 1003 + 17 * rank + 1000 * i, dev) for i in range(nbuf)]
    if os.environ.get("BFA_BENCH_SAME_INPUT"):  # (experiment: every utterance reads utterance 0 -> cache-resident rows)
        bufs = [(lp[:1].expand(B, T, C), tk[:1].expand(B, S).contiguous()) for lp, tk in bufs]
    T_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    S_len = torch.full((B,), S, dtype=torch.int32, device=dev)
    # reference defaults: anchors 10, boost, floor, truly_forced.  One decoder (= one workspace) per batch in flight.
    from bournemouth_forced_aligner_amd import BatchesInFlight
    # the product's own helper (inflight.py); the batches are resident and complete before the timed region starts
    bif = BatchesInFlight(blank, sil, n=max(1, inflight, npipe), device=dev,
                          wait_for_caller=bool(int(os.environ.get("BFA_BENCH_WAIT_CALLER", "0"))))
    aus = bif.decoders
    au = aus[0]
    lib = _lib.lib()
    hs = [_lib.handle(rk.local_rank, k) for k in range(len(aus))]
    # (a stream of its own priority gets a hardware queue of its own; streams of one priority share a few queues, and a tail
    # stream that lands on the queue of the launch stream would run in submission order behind the next K1)
    tail_prio = int(os.environ.get("BFA_BENCH_TAIL_PRIO", "-1"))
    tail = torch.cuda.Stream(device=dev, priority=tail_prio) if npipe > 1 else None

    # the lengths are known on the host (constant here), and the synthetic targets never contain SIL:
    # tell the library which K1 register class occurs so that it does not launch the empty ones
    hint = au.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=(None if args.no_window else C))

    streams = bif.streams if (inflight > 1 and npipe == 1) else None

    def step(i):
        lp, tk = bufs[i % nbuf]
        if tail is not None:
            return aus[i % npipe].decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint, tail_stream=tail)
        if streams is None:
            return au.decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint)
        # batches i, i+1, i+2 on different streams with their own decoders: the latency-bound tail of one (backtrace,
        # run-length encoding) and the ramp-down of its K1 overlap the K1 of the next ones
        return bif.submit(lp, tk, T_len, S_len, class_mask=hint)

    res = None
    for i in range(args.warmup):
        res = step(i)
    torch.cuda.synchronize()

    def check_status(r):
        if r is not None and not os.environ.get("BFA_HIP_LIBRARY"):  # (kernel-time experiments with a stubbed role produce garbage)
            st = r.status.cpu().numpy()
            assert (st == 0).all(), f"alignment failed on the bench workload: {np.unique(st)}"
    check_status(res)

    # K1 of every step is bracketed with HIP events on the launch stream (measured: no effect on the step time;
    # BFA_BENCH_K1_EVERY=n samples every n-th step instead)
    k1_every = int(os.environ.get("BFA_BENCH_K1_EVERY", "1"))

    def timed_window():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; K1 events collected."""
        for h in hs:
            lib.bfa_profile_enable(h, k1_every)
        rk.barrier()
        torch.cuda.synchronize()
        base = torch.cuda.Event(enable_timing=True)
        base.record()
        t0 = time.perf_counter()
        r = None
        for i in range(args.steps):
            r = step(i)
        issue = time.perf_counter() - t0  # host time to enqueue the steps (must stay below the device time per step)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        rk.barrier()
        el = time.perf_counter() - t0
        spans = []
        for h in hs:
            lib.bfa_profile_enable(h, 0)
            a0 = (ctypes.c_float * max(1, args.steps))()
            a1 = (ctypes.c_float * max(1, args.steps))()
            n = lib.bfa_profile_collect_spans(h, ctypes.c_void_p(base.cuda_event), a0, a1, args.steps)
            spans += [(float(a0[i]), float(a1[i])) for i in range(n)]
        spans.sort()
        return el, mine, [b - a for a, b in spans], r, spans, issue

    # Window 1: W warm-up steps, then K steps -- the first milliseconds of load.  On MI355X the power management
    # reacts to the load step: K1 starts at its steady duration, rises by ~15 % between ~2 and ~15 ms after the start
    # and settles back after ~30-50 ms (profiles/r02_k1_series.txt); a 20-step window right after 5 warm-up steps lands
    # exactly in that transient.  So the steps are then kept running, untimed, for --settle-ms, and window 2 (again
    # EXACTLY K steps, same barriers) measures the settled state.  `value` is window 2; window 1 is reported beside it.
    first = timed_window()
    settle_steps = 0
    if args.settle_ms > 0:
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        while (time.perf_counter() - s0) * 1e3 < args.settle_ms:
            for i in range(8):
                step(settle_steps + i)
            settle_steps += 8
            torch.cuda.synchronize()
        elapsed, mine, k1s, res, spans, issue = timed_window()
    else:
        elapsed, mine, k1s, res, spans, issue = first
    check_status(res)
    nk1 = len(k1s)
    k1_ms = float(np.mean(k1s)) if nk1 > 0 else float("nan")
    if os.environ.get("BFA_BENCH_DUMP_K1"):
        print("k1 series (ms), first window:", " ".join(f"{v:.3f}" for v in first[2]), file=sys.stderr)
        print("k1 series (ms), settled window:", " ".join(f"{v:.3f}" for v in k1s), file=sys.stderr)
    elapsed, _ = rk.max_over_ranks(elapsed)
    first_elapsed, _ = rk.max_over_ranks(first[0])
    _, rank_ms = rk.max_over_ranks(mine / args.steps * 1e3)

    # confidence pass (utils._calculate_confidences), timed separately: it is a separate reference call
    lp0, _ = bufs[(args.steps - 1) % nbuf]
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(5):
        conf, _st = calculate_confidences_batch(lp0, res.segs, res.seg_count)
    torch.cuda.synchronize()
    conf_ms = (time.perf_counter() - c0) / 5 * 1e3

    # final gather of the (small) result records over RCCL, outside the timed steps
    gather_ms = None
    if dist is not None:
        gidx = torch.arange(rank * B, (rank + 1) * B, dtype=torch.int64)
        gather_results(res.segs, res.seg_count, conf, gidx, world * B)  # warm (communicator set-up)
        torch.cuda.synchronize()
        rk.barrier()
        g0 = time.perf_counter()
        out = gather_results(res.segs, res.seg_count, conf, gidx, world * B)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            gs, gc, _gf = out
            assert gs.shape[0] == world * B and torch.equal(gs[:B, :res.segs.shape[1]], res.segs) \
                and torch.equal(gc[:B], res.seg_count), "gathered records differ from rank 0's own results"

    frames_per_step = B * T
    L = 4 * S + 1
    bytes_per_frame = 4 * C + (L + 3) // 4 + 8
    value = world * frames_per_step * args.steps / elapsed
    # time during which at least one K1 launch was running (union of the launch intervals), per launch
    busy_ms, cur_a, cur_b = 0.0, None, None
    for a_, b_ in spans:
        if cur_b is None or a_ > cur_b:
            busy_ms += (cur_b - cur_a) if cur_b is not None else 0.0
            cur_a, cur_b = a_, b_
        else:
            cur_b = max(cur_b, b_)
    busy_ms += (cur_b - cur_a) if cur_b is not None else 0.0
    busy_per_launch = busy_ms / nk1 if nk1 > 0 else float("nan")
    achieved = frames_per_step * bytes_per_frame / (busy_per_launch * 1e-3) / 1e9 if nk1 > 0 else None
    achieved_launch = frames_per_step * bytes_per_frame / (k1_ms * 1e-3) / 1e9 if k1_ms == k1_ms else None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:  # timed at N=1 only
        cpu = cpu_baseline_leg(args, bufs, step, B, T, S, C, blank, sil)

    # HBM bytes of one K1 launch from the PMC passes of tools/profile.sh (committed under profiles/), if it was
    # taken on this workload; FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes
    traffic, tfile = None, None
    for name in ("r02_k1_traffic.json", "r01_k1_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath) and (B, T, S, C) == (4096, 1000, 40, 67):
            traffic, tfile = json.load(open(tpath))["traffic_bytes_per_launch"], name
            break

    ranks = rk.describe()
    if rank == 0:
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors", "value": value,
            "unit": "aligned frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch={B} T={T} |tokens|={S} ph66 (C={C}) per GPU, reference-default flags "
                                   f"(boost+floor+truly_forced, anchors=10, no SIL in targets -> standard mode)",
                       "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, no data-path collective",
                       "batches_in_flight": max(1, inflight),
                       "pipeline": (f"{npipe} decoders taking turns on one stream, walk / run-length encoding of a step on a "
                                    f"shared tail stream beside the K1 of the next (bfa_set_tail_stream)") if npipe > 1
                       else "none (stream-ordered calls)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": f"bytes per K1 launch (rocprofv3 PMC, profiles/{tfile})" if tfile else None,
                         "algorithmic_bytes_per_launch": frames_per_step * bytes_per_frame,
                         "kernel": "k_dp4w<2,4,3> (K1 banded Viterbi forward, sliding-window consumer)",
                         "what": "achieved = algorithmic bytes of a K1 launch / K1 busy time per launch; busy time = union of "
                                 "the K1 launch intervals in the timed region (HIP events of the library on the launch "
                                 "streams) -- with one batch in flight it IS the mean launch duration, with several the "
                                 "launches overlap and share the machine",
                         "kernel_busy_ms_per_launch": busy_per_launch, "kernel_busy_ms": busy_ms,
                         "launches_running_on_average": (sum(k1s) / busy_ms) if busy_ms > 0 else None,
                         "kernel_ms": k1_ms, "kernel_ms_stats": _stats(k1s), "kernel_ms_samples": int(nk1),
                         "achieved_per_launch_duration": achieved_launch,
                         "frac_per_launch_duration": (achieved_launch / HBM_PEAK_GBS) if achieved_launch else None,
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "whole_step_frac": frames_per_step * bytes_per_frame / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
            "cpu_baseline": cpu,
            "reference_cpu_baseline": _reference_cpu_record(),
            "settle": {"settle_ms": args.settle_ms, "untimed_steps_between_windows": settle_steps,
                       "first_window": {"what": f"the {args.steps} steps right after the {args.warmup} warm-up steps "
                                                "(power-management transient of the first ~30 ms of load)",
                                        "ms_per_step": first_elapsed / args.steps * 1e3,
                                        "value": world * frames_per_step * args.steps / first_elapsed,
                                        "kernel_ms_stats": _stats(first[2])}},
            "host_issue_ms_per_step": issue / args.steps * 1e3,
            "confidence_pass_ms": conf_ms,
            "gather_ms": gather_ms,
            "rank_ms_per_step": rank_ms,
            "ranks": ranks,
        }
        print(json.dumps(line))


def cpu_baseline_leg(args, bufs, step, B, T, S, C, blank, sil):
    from oracle import oracle as ora
    prm = ora.make_params(blank, sil)
    per = min(args.cpu_sample, B)
    nbuf = max(1, min(2, (args.cpu_sample + B - 1) // B))
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
    return cpu


# --------------------------------------------------------------------------------------------------- C4
def _shard_digest(idx):
    """order-sensitive 63-bit digest of an index array (every rank must hold the same partition)"""
    import hashlib
    return int.from_bytes(hashlib.sha256(np.ascontiguousarray(idx, np.int64).tobytes()).digest()[:8], "little") >> 1


def c4_plan(n_total, world, seed, chunk):
    """Host-side plan, identical on every rank: lengths, the LPT partition (sharding.shard_utterances on T*(4S+1)),
    and per shard the length-sorted sub-batches ("chunks") one bfa_align_batch call takes."""
    from bournemouth_forced_aligner_amd.sharding import shard_utterances, utterance_cost
    T, S = c4_lengths(n_total, seed)
    shards = shard_utterances(T, S, world)
    plans = []
    for s in shards:
        order = s[np.argsort(-T[s], kind="stable")]  # longest first: similar lengths share a call (less padding,
        plans.append([order[i:i + chunk] for i in range(0, len(order), chunk)])  # fewer K1 classes per call)
    cost = utterance_cost(T, S)
    loads = np.array([cost[s].sum() for s in shards], np.float64)
    return T, S, shards, plans, loads


def c4_main(args, rk):
    dev, rank, world, dist = rk.dev, rk.rank, rk.world, rk.dist
    C, seed, n_total = args.classes, args.seed, args.global_batch
    blank = C - 1
    T, S, shards, plans, loads = c4_plan(n_total, world, seed, args.chunk)
    mine = shards[rank]
    # every rank must have computed the same partition
    digest = _shard_digest(np.concatenate([np.asarray([len(x) for x in shards], np.int64)] + list(shards)))
    _, digests = rk.max_over_ranks(float(digest % (1 << 52)))
    shard_agree = len(set(digests)) == 1
    assert shard_agree, f"ranks disagree on the shard partition: {digests}"
    allidx = np.sort(np.concatenate(shards))
    assert np.array_equal(allidx, np.arange(n_total)), "the shards are not a partition of the batch"

    from bournemouth_forced_aligner_amd.sharding import gather_results
    cap = int(S.max()) + 2
    if rk.dry:
        # no GPU work: exercise launch, partition agreement and the gather plumbing with fabricated records
        segs = torch.zeros((len(mine), cap, 4), dtype=torch.int32)
        segs[:, 0, 0] = torch.from_numpy(mine).to(torch.int32)
        cnt = torch.from_numpy(np.minimum(S[mine], cap)).to(torch.int32)
        out = gather_results(segs, cnt, None, torch.from_numpy(mine), n_total)
        ok = None
        if rank == 0:
            gs, gc, _ = out
            ok = bool(torch.equal(gs[:, 0, 0].long(), torch.arange(n_total)) and
                      np.array_equal(gc.numpy(), np.minimum(S, cap)))
        ranks = rk.describe()
        if rank == 0:
            print(json.dumps({"dry_run": True, "config": "c4", "n_gpus": world, "global_batch": n_total,
                              "shard_sizes": [int(len(s)) for s in shards], "shard_digest": digest,
                              "shard_agree": shard_agree, "load_imbalance_max_over_mean": float(loads.max() / loads.mean()),
                              "gather_ok": ok, "ranks": ranks}))
        return

    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    au = AlignmentUtils(blank_id=blank, silence_id=0)
    vd = au.viterbi_decoder
    lib = _lib.lib()
    h = _lib.handle(rk.local_rank)

    # ---- synthesis: each rank makes only its own utterances (sub-batches of <= 512 to bound the temporaries)
    t_s0 = time.perf_counter()
    chunks = []
    for ch in plans[rank]:
        Tp, Sp = int(T[ch].max()), int(S[ch].max())
        lps, tks = [], []
        for i in range(0, len(ch), 512):
            sub = ch[i:i + 512]
            lp, tk = c4_utterances(sub, T[sub], S[sub], C, seed, dev, Tpad=Tp, Spad=Sp)
            lps.append(lp)
            tks.append(tk)
        lp = torch.cat(lps, 0) if len(lps) > 1 else lps[0]
        tk = torch.cat(tks, 0) if len(tks) > 1 else tks[0]
        del lps, tks
        Td = torch.from_numpy(T[ch].astype(np.int32)).to(dev)
        Sd = torch.from_numpy(S[ch].astype(np.int32)).to(dev)
        hint = vd.class_mask_hint(T[ch], S[ch], has_sil=False, n_classes=C)
        chunks.append(dict(idx=ch, lp=lp, tk=tk, Td=Td, Sd=Sd, hint=hint, csum=input_checksum(lp, T[ch])))
    torch.cuda.synchronize()
    synth_s = time.perf_counter() - t_s0
    my_frames = int(T[mine].sum())
    my_bytes = int(((4 * C + (4 * S[mine] + 1 + 3) // 4 + 8) * T[mine]).sum())

    # --inflight k: k steps in flight, each on its own stream with its own decoder (workspace) and library handle (aux
    # streams).  A rank's shard of a sharded batch is bound by the chain of its longest utterance, not by the machine;
    # a service that aligns a stream of such batches overlaps them.  `value` stays frames / wall time.
    nfl = max(1, args.inflight or 1)
    aus = [au]
    for k in range(1, nfl):
        a2 = AlignmentUtils(blank_id=blank, silence_id=0)
        a2.viterbi_decoder.handle_slot = k
        aus.append(a2)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)] if nfl > 1 else None
    step_no = [0]

    def step():
        k = step_no[0] % nfl
        step_no[0] += 1
        if streams is None:
            return [au.decode_alignments_device(c["lp"], c["tk"], c["Td"], c["Sd"], class_mask=c["hint"], seg_cap=cap)
                    for c in chunks]
        with torch.cuda.stream(streams[k]):
            return [aus[k].decode_alignments_device(c["lp"], c["tk"], c["Td"], c["Sd"], class_mask=c["hint"], seg_cap=cap)
                    for c in chunks]

    for _ in range(max(1, args.warmup)):
        res = step()
    torch.cuda.synchronize()
    for r in res:
        st = r.status.cpu().numpy()
        assert (st == 0).all(), f"alignment failed on the C4 workload: {np.unique(st)}"

    lib.bfa_profile_enable(h, 1)
    rk.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    rk.barrier()
    elapsed = time.perf_counter() - t0
    lib.bfa_profile_enable(h, 0)
    ncall = args.steps * len(chunks)
    k1 = (ctypes.c_float * max(1, ncall))()
    nk1 = lib.bfa_profile_collect(h, k1, ncall)
    k1_step_ms = float(np.sum([k1[i] for i in range(nk1)])) / args.steps if nk1 else float("nan")
    elapsed, _ = rk.max_over_ranks(elapsed)
    _, rank_ms = rk.max_over_ranks(mine_s / args.steps * 1e3)

    # ---- final gather of the result records (outside the steps; timed on its own, after a warm-up gather that
    # pays for the communicator set-up)
    segs = torch.cat([r.segs for r in res], 0)
    cnt = torch.cat([r.seg_count for r in res], 0)
    gidx = torch.from_numpy(np.concatenate([c["idx"] for c in chunks]))
    csum = torch.cat([c["csum"] for c in chunks], 0)
    # the input checksum rides along as two int32 "segments" in a spare row (cap + 1 rows)
    segs_x = torch.zeros((segs.shape[0], cap + 1, 4), dtype=torch.int32, device=dev)
    segs_x[:, :cap] = segs
    cs62 = csum & ((1 << 62) - 1)
    segs_x[:, cap, 0] = (cs62 & ((1 << 31) - 1)).to(torch.int32)
    segs_x[:, cap, 1] = (cs62 >> 31).to(torch.int32)
    gather_results(segs_x, cnt, None, gidx, n_total)
    torch.cuda.synchronize()
    rk.barrier()
    g0 = time.perf_counter()
    out = gather_results(segs_x, cnt, None, gidx, n_total)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - g0) * 1e3
    gather_ms, _ = rk.max_over_ranks(gather_ms)

    total_frames = int(T.sum())
    total_bytes = int(((4 * C + (4 * S + 1 + 3) // 4 + 8) * T).sum())
    step_s = elapsed / args.steps
    parity = None
    if rank == 0:
        gs, gc, _ = out
        parity = c4_parity_sample(args, T, S, gs, gc, cap, dev, C, seed)
    ranks = rk.describe()
    if rank == 0:
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors", "value": total_frames / step_s,
            "unit": "aligned frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: global batch={n_total} mixed-length T~U[200,3000] S=T//25 ph66 (C={C}), seed {seed}, "
                                   f"reference-default flags; {nfl} step(s) in flight; LPT-sharded over {world} rank(s), "
                                   f"{len(chunks)} length-sorted calls of <= {args.chunk} utterances per rank",
                       "global_batch": n_total, "parallelism": f"utterance-sharded x{world} (LPT on T*(4S+1)), "
                                                               f"no data-path collective, final gather of records"},
            "frames_per_step": total_frames,
            "value_with_gather": total_frames / (step_s + gather_ms * 1e-3),
            "gather_ms": gather_ms,
            "rank_ms_per_step": rank_ms,
            "rank_ms_max_over_mean": float(max(rank_ms) / np.mean(rank_ms)),
            "planned_load_max_over_mean": float(loads.max() / loads.mean()),
            "shard_sizes": [int(len(s)) for s in shards], "shard_agree": shard_agree,
            "synthesis_s_rank0": synth_s,
            "roofline": {"bound": "hbm", "achieved": total_bytes / step_s / 1e9 / world, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s per GPU", "frac": total_bytes / step_s / 1e9 / world / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step (the K1 class kernels of a call run side by side; K1 span per step "
                                   "from HIP events on rank 0 in kernel_ms)",
                         "kernel_ms": k1_step_ms, "algorithmic_bytes_per_step": total_bytes,
                         "rank0_frames": my_frames, "rank0_algorithmic_bytes": my_bytes},
            "parity_sample": parity,
            "ranks": ranks,
        }
        print(json.dumps(line))


def c4_parity_sample(args, T, S, gs, gc, cap, dev, C, seed):
    """Rank 0, after the gather: a stratified sample (every length stratum + the longest utterances) is synthesised
    again from the global indices, its input checksum compared with the one the owning rank gathered, and the
    gathered records compared with the oracle's."""
    from oracle import oracle as ora
    n_total = len(T)
    n = min(args.parity_sample, n_total)
    order = np.argsort(T, kind="stable")
    longest = order[-min(16, n):]
    strat = order[np.linspace(0, n_total - 1, max(0, n - len(longest))).astype(np.int64)]
    sample = np.unique(np.concatenate([longest, strat]))
    prm = ora.make_params(C - 1, 0)
    mism, bad_inputs, frames = 0, 0, 0
    w = 0.0
    gs_h = gs.cpu().numpy()
    gc_h = gc.cpu().numpy()
    for i in range(0, len(sample), 64):
        sub = sample[i:i + 64]
        lp, tk = c4_utterances(sub, T[sub], S[sub], C, seed, dev)
        cs = input_checksum(lp, T[sub]).cpu().numpy()
        lp_h, tk_h = lp.cpu().numpy(), tk.cpu().numpy()
        w0 = time.perf_counter()
        exp = ora.decode_alignments(lp_h, tk_h, T[sub], S[sub], prm, seg_cap=cap)
        w += time.perf_counter() - w0
        for k, g in enumerate(sub):
            frames += int(T[g])
            got_cs = int(gs_h[g, cap, 0]) + (int(gs_h[g, cap, 1]) << 31)
            if got_cs != int(cs[k]) & ((1 << 62) - 1):
                bad_inputs += 1
            c = int(exp["seg_count"][k])
            if int(gc_h[g]) != c or not (gs_h[g, :c] == exp["seg"][k, :c]).all():
                mism += 1
    return {"utterances": int(len(sample)), "frames": frames, "longest_T": int(T[sample].max()),
            "mismatching_utterances": mism, "regenerated_inputs_differing": bad_inputs,
            "oracle_s": w, "oracle_frames_per_s_1core": frames / w if w > 0 else None}


# ----------------------------------------------------------------------------------------------- ragged
def ragged_main(args, rk):
    """Side measurement: throughput on ONE unsorted mixed-length call (prints its own JSON; its parity check against
    the oracle is tests/test_gpu_parity.py::test_mixed_length_shard_parity)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    dev, rank = rk.dev, rk.rank
    C, B = args.classes, args.batch
    lp, tk, T_len, S_len = synth_ragged(B, args.tlo, args.thi, C, 1004 + rank, dev)
    au = AlignmentUtils(blank_id=C - 1, silence_id=0)
    au.viterbi_decoder.window_max_frames = args.win_frames or None
    au.viterbi_decoder.window_max_tokens = args.win_tokens or None
    hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False,
                                              n_classes=(None if args.no_window else C))
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
    S_np, T_np = S_len.numpy().astype(np.int64), T_len.numpy().astype(np.int64)
    nbytes = int(((4 * C + (4 * S_np + 1 + 3) // 4 + 8) * T_np).sum())
    if rank == 0:
        print(json.dumps({"workload": f"ragged batch={B} T~U[{args.tlo},{args.thi}] S=T//25 C={C}", "frames": frames,
                          "ms_per_step": el * 1e3, "frames_per_s": frames / el,
                          "algorithmic_bytes": nbytes, "hbm_frac": nbytes / el / 1e9 / HBM_PEAK_GBS,
                          "status_ok": bool((res.status.cpu() == 0).all())}))


def dry_headline(args, rk):
    """--dry-run of the headline mode: launch + gather plumbing only."""
    from bournemouth_forced_aligner_amd.sharding import gather_results
    B, cap = 64, args.tokens + 2
    segs = torch.zeros((B, cap, 4), dtype=torch.int32)
    gidx = torch.arange(rk.rank * B, (rk.rank + 1) * B, dtype=torch.int64)
    segs[:, 0, 0] = gidx.to(torch.int32)
    cnt = torch.full((B,), 1, dtype=torch.int32)
    if rk.dist is not None:
        out = gather_results(segs, cnt, None, gidx, rk.world * B)
    else:
        out = (segs, cnt, None)
    ranks = rk.describe()
    if rk.rank == 0:
        ok = bool(torch.equal(out[0][:, 0, 0].long(), torch.arange(rk.world * B)))
        print(json.dumps({"dry_run": True, "config": "headline", "n_gpus": rk.world, "gather_ok": ok, "ranks": ranks}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["headline", "c4"], default="headline")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="headline, experiment (measured: slower than plain calls): n decoders taking turns on one stream "
                         "with a shared tail stream (bfa_set_tail_stream); 1 = off")
    ap.add_argument("--tlo", type=int, default=200, help="--ragged: shortest utterance")
    ap.add_argument("--thi", type=int, default=3000, help="--ragged: longest utterance")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--tokens", type=int, default=40)
    ap.add_argument("--classes", type=int, default=67)
    ap.add_argument("--settle-ms", type=float, default=120.0,
                    help="headline: untimed steps run for this long between the first and the reported window (0 = report "
                         "the first window)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight, each on its own stream with its own decoder / library handle / workspace "
                         "(default: 3 for the headline, 1 for --config c4)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="utterances timed on the host oracle (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-window", action="store_true", help="A/B: full state layout instead of the sliding window")
    ap.add_argument("--win-frames", type=int, default=0, help="A/B: window frame limit, 0 = default")
    ap.add_argument("--win-tokens", type=int, default=0, help="A/B: window token limit, 0 = default")
    ap.add_argument("--ragged", action="store_true",
                    help="side measurement: ONE unsorted mixed-length call T~U{200..3000}, S=T//25")
    ap.add_argument("--global-batch", type=int, default=32768, help="c4: utterances over all ranks")
    ap.add_argument("--chunk", type=int, default=16384, help="c4: utterances per bfa_align_batch call (one call per rank when the shard is smaller)")
    ap.add_argument("--seed", type=int, default=1004, help="c4: generator seed")
    ap.add_argument("--parity-sample", type=int, default=256, help="c4: utterances rank 0 checks against the oracle")
    ap.add_argument("--dry-run", action="store_true", help="launch + partition + gather plumbing on gloo, no GPU work")
    ap.add_argument("--force-group", action="store_true",
                    help="initialise the process group even at world size 1 (exercises the N > 1 code path of the headline mode)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))

    rk = Ranks(args, need_group=(args.config == "c4" or args.force_group))
    if args.ragged:
        ragged_main(args, rk)
    elif args.config == "c4":
        c4_main(args, rk)
    elif args.dry_run:
        dry_headline(args, rk)
    else:
        headline_main(args, rk)
    rk.close()  # (not in a `finally`: a rank that failed must not wait in the closing barrier)


if __name__ == "__main__":
    main()
