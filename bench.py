#!/usr/bin/env python3
"""bench.py -- aligned frames/s of the MI355X forced-alignment core on synthetic ph66 posteriors.

A "step" is one pass of the hot path (bfa_align_batch: planning + K1 banded Viterbi forward with
fused boost/log_softmax/floor + K2 backtrace + run-length encoding) over one batch of device-resident
log-probabilities.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      headline: BASELINE.json configs[2] ("batch=4096 T=1000 |tokens|=40 ph66") PER GPU, weak scaling.  The K timed
      steps are issued with four batches in flight (step i on stream i % 4 with its own decoder / library handle /
      workspace / outputs; --inflight 1 = one batch at a time): the latency-bound tail of a step runs beside the forward
      kernel of the next ones.  A step is alignment (K1), walk / tuples (K2) AND the confidence pass of the aligned tuples
      (K3, core.py:936-937) unless --no-confidences; `alignment_only` in the line is the K1 + K2 step of rounds 1-3.
      Every step does all of its work inside the timed region; value = frames / wall time.
  python bench.py --config c4 [--gpus N]
      BASELINE.json configs[3]: global B = 32768, T in [200,3000], S = T // 25 (seed 1004), LPT-sharded over the N
      ranks (sharding.shard_utterances), every rank synthesises and aligns only its shard, the result records are
      gathered to rank 0 (sharding.gather_results, RCCL) and a stratified sample is checked against the oracle there.

Ranks: one process per GPU.  Under `torch.distributed.run` (what the driver does for N > 1) the ranks come from the
environment; started directly with --gpus N > 1 the script re-executes itself under torch.distributed.run with N
processes.  `n_gpus` in the JSON line is the world size the process group reports, `ranks` lists the device of each.
`--dry-run` does the same launch with the gloo backend and no GPU work (partition + gather plumbing only; CPU test).

  python bench.py --config realtext
      the path real transcripts take: raw logits of both heads, SIL in the targets (silence-anchored mode),
      bfa_align_heads + bfa_postprocess + bfa_confidences per step; 512-utterance oracle parity count in the line.

  python bench.py --config c5proxy [--peak P]
      the stand-in for BASELINE.json configs[4] (the real model's checkpoint is not available offline): mixed-length segments
      of <= 30 s, S = T // 12, both heads from raw logits, SIL in the targets, one bfa_align_heads call per step.
  --peak P / --sigma S (every config): sharpness of the synthetic posteriors, logits = N(0, S) + P * onehot(planted); the line
      carries `softness` (log-probability per frame of the aligned path, share of utterances at the reference's -1000
      sentinel, what the fast windows did).  tools/softness.py sweeps it over the call shapes.

Timing protocol: W warm-up steps, then ceil(100 / K) windows of EXACTLY K steps, each bracketed by barrier +
synchronize on both sides, max over ranks; `value` = frames of all windows / sum of the window times.  Everything that
ran before the reported windows (warm-up, the first window -- reported separately --, the --settle-ms steps) is counted
in `timing.warmup_effective_steps`.

One JSON line is printed by rank 0.  `roofline` prices the dominant kernel (K1) with the algorithmic bytes of
SURVEY.md section 8(d): 4*C + ceil(L/4) + 8 bytes per frame, against K1's mean launch duration in a leg of the same
run with ONE batch in flight (no launch overlaps another), which is what `rocprofv3 --kernel-trace --stats` of
`python bench.py --inflight 1` reports for the kernel.  `cpu_baseline` times the C restatement of the reference
(oracle/, kind "port") on the host (N = 1 only).
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
        # --oversubscribe: N ranks doing REAL GPU work on fewer GPUs than ranks (device = LOCAL_RANK % device_count) with a
        # host-side (gloo) process group: RCCL refuses two ranks on one device.  Timing means nothing there; what it proves is
        # partition -> synthesis -> alignment -> pack -> gather -> index -> oracle check with N > 1 ranks of real results.
        self.over = bool(getattr(args, "oversubscribe", False)) and not self.dry
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU implementation of the path)"
            assert self.over or self.local_rank < torch.cuda.device_count(), \
                f"rank {self.rank}: local rank {self.local_rank} but only {torch.cuda.device_count()} GPUs are visible"
            self.dev = torch.device("cuda", self.local_rank % torch.cuda.device_count())
            torch.cuda.set_device(self.dev)
        self.coll_dev = torch.device("cpu") if (self.dry or self.over) else self.dev   # where collective operands live
        self.backend, self.ranks_seen = None, None
        if self.world > 1 or need_group:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()) if self.world == 1 else "29500")
            if self.dry or self.over:
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dist = dist
            self.world = dist.get_world_size()  # what the backend reports
            # how many ranks the communicator itself saw: an all_reduce of ones over the group (RCCL with backend nccl)
            ones = torch.ones(1, dtype=torch.int32, device=self.coll_dev)
            dist.all_reduce(ones)
            self.backend, self.ranks_seen = dist.get_backend(), int(ones.item())
        if args.gpus != self.world and self.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the process group has {self.world} ranks; reporting {self.world}",
                  file=sys.stderr)

    def describe(self):
        """[{rank, device, name, ...}] for every rank (gathered), so the line shows which GPUs really took part."""
        if self.dry:
            me = {"rank": self.rank, "device": "cpu (dry run)", "pid": os.getpid(), "local_rank": self.local_rank,
                  "would_pin": f"cuda:{self.local_rank}", "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        else:
            pr = torch.cuda.get_device_properties(self.dev)
            me = {"rank": self.rank, "device": f"cuda:{self.dev.index}", "local_rank": self.local_rank,
                  "current_device": int(torch.cuda.current_device()), "name": pr.name,
                  "gcn_arch": getattr(pr, "gcnArchName", ""), "cus": pr.multi_processor_count,
                  "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")),
                  "hbm_gb": round(pr.total_memory / 2 ** 30, 1), "pid": os.getpid()}
            if self.over:
                me["device"] = f"cuda:{self.dev.index}"
                me["oversubscribed"] = True
        if self.dist is None:
            return [me]
        out = [None] * self.world
        self.dist.all_gather_object(out, me)
        return out

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def group_record(self):
        """{"backend", "ranks_seen", "rccl_ranks_seen"}: what the communicator saw (all_reduce of ones at start-up)"""
        return {"backend": self.backend, "ranks_seen": self.ranks_seen,
                "rccl_ranks_seen": self.ranks_seen if self.backend == "nccl" else None,
                "oversubscribed": self.over}

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x), [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device=self.coll_dev)
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
def _timed_windows(rk, run_steps, K, n_windows):
    """n_windows windows of EXACTLY K steps each, every window bracketed by barrier + synchronize on both sides; the
    window time is the max over ranks.  Returns (list of window seconds, list of host issue seconds)."""
    times, issues = [], []
    for _ in range(n_windows):
        rk.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(K)
        issues.append(time.perf_counter() - t0)  # host time to enqueue the steps (must stay below the device time)
        torch.cuda.synchronize()
        rk.barrier()
        el, _ = rk.max_over_ranks(time.perf_counter() - t0)
        times.append(el)
    return times, issues


def _union_ms(spans):
    """total length of the union of (start, end) intervals"""
    busy, cur_a, cur_b = 0.0, None, None
    for a_, b_ in sorted(spans):
        if cur_b is None or a_ > cur_b:
            busy += (cur_b - cur_a) if cur_b is not None else 0.0
            cur_a, cur_b = a_, b_
        else:
            cur_b = max(cur_b, b_)
    return busy + ((cur_b - cur_a) if cur_b is not None else 0.0)


def measured_copy_peak(dev, lib, h, nbytes=1 << 30, reps=12):
    """The copy ceiling of THIS GPU in this process: a float4 copy kernel of the library (bfa_profile_copy)
    over 1 GiB (2 GiB of HBM traffic per launch), HIP events on the launch stream; the best and the mean of `reps` launches."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.random_(0, 255)
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        lib.bfa_profile_copy(h, dst.data_ptr(), src.data_ptr(), nbytes, st.cuda_stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        lib.bfa_profile_copy(h, dst.data_ptr(), src.data_ptr(), nbytes, st.cuda_stream)
        e1.record(st)
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(src[:1 << 20], dst[:1 << 20]) and torch.equal(src[-(1 << 20):], dst[-(1 << 20):]))
    del src, dst
    return {"gbs_best": 2 * nbytes / (min(ms) * 1e-3) / 1e9, "gbs_mean": 2 * nbytes / (float(np.mean(ms)) * 1e-3) / 1e9,
            "bytes_moved_per_launch": 2 * nbytes, "launches": reps, "copied_correctly": ok}


def k1_valu_from_pmc(kernel_prefix="k_dp4w<2, 4, 3"):
    """SQ_INSTS_VALU of the headline K1 per launch from the newest committed PMC summary (tools/pmc.sh format)."""
    for name in ("r06_headline_pmc.txt", "r05_headline_pmc.txt", "r04_headline_pmc.txt"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        cur = None
        for ln in open(path):
            if not ln.startswith(" "):
                cur = ln.strip()
            elif cur and cur.startswith(kernel_prefix) and ln.split()[0] == "SQ_INSTS_VALU":
                return float(ln.split()[1]), name
    return None, None


N_SIMD, SIMD_CLOCK_HZ = 1024, 2.4e9   # MI355X: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles


def headline_main(args, rk):
    from bournemouth_forced_aligner_amd import BatchesInFlight, calculate_confidences_batch, _lib
    from bournemouth_forced_aligner_amd.sharding import gather_results
    dev, rank, world, dist = rk.dev, rk.rank, rk.world, rk.dist
    B, T, S, C = args.batch, args.frames, args.tokens, args.classes
    blank, sil = C - 1, 0
    K = args.steps
    # Batches in flight (default 4; 3 until round 4, when the confidence pass joined the step: 0.407 against 0.423 ms on one box,
    # tools/experiments/scripts/r4_inflight.sh): step i runs on stream i % n with decoder i % n (own library handle, workspace and
    # outputs), so the tail of a step (rerun launch, walk, run-length encoding: ~65 us of latency chains that leave the
    # machine mostly idle) and the ramp-down of its K1 overlap the K1 of the next steps.  `value` = frames / wall time.
    # The K1 launches of different steps then overlap each other, so a launch's duration there says how long it SHARED
    # the machine: the roofline entry is priced by a separate leg of this same run with ONE batch in flight (below).
    inflight = max(1, args.inflight if args.inflight is not None else 4)
    nbuf = max(2, inflight)
    # distinct batches so consecutive steps never stream the same 1.1 GB (> the 256 MB Infinity Cache anyway)
    bufs = [synth_batch(B, T, S, C, 1003 + 17 * rank + 1000 * i, dev, peak=args.peak, sigma=args.sigma) for i in range(nbuf)]
    if os.environ.get("BFA_BENCH_SAME_INPUT"):  # (experiment: every utterance reads utterance 0 -> cache-resident rows)
        bufs = [(lp[:1].expand(B, T, C), tk[:1].expand(B, S).contiguous()) for lp, tk in bufs]
    if args.row_pitch:  # A/B: the same log-probs in rows padded to `row_pitch` floats (e.g. 72 = 288 B)
        assert args.row_pitch >= C
        padded = []
        for lp, tk in bufs:
            wide = torch.zeros((B, T, args.row_pitch), dtype=torch.float32, device=dev)
            wide[:, :, :C] = lp
            padded.append((wide[:, :, :C], tk))  # a [B,T,C] view with row stride `row_pitch`
        bufs = padded
    if args.place_align:  # A/B: the posteriors at a chosen address alignment (+ offset) instead of wherever the caching allocator put them
        placed = []
        for lp, tk in bufs:
            nbytes = lp.numel() * 4
            raw = torch.empty(nbytes + args.place_align + args.place_offset + 512, dtype=torch.uint8, device=dev)
            off = (-raw.data_ptr()) % args.place_align + args.place_offset
            view = raw[off:off + nbytes].view(torch.float32).view(lp.shape)
            view.copy_(lp)
            placed.append((view, tk))
        bufs = placed
    if os.environ.get("BFA_BENCH_DUMP_K1"):
        print("posterior buffers: " + " ".join(f"0x{lp.data_ptr():x} (mod 2 MiB = {lp.data_ptr() % (1 << 21)})" for lp, _ in bufs), file=sys.stderr)
    T_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    S_len = torch.full((B,), S, dtype=torch.int32, device=dev)
    # the product's own helper (inflight.py): one decoder (= library handle, workspace, outputs) per batch in flight;
    # reference defaults: anchors 10, boost, floor, truly_forced.  The batches are resident before the timed region.
    bif = BatchesInFlight(blank, sil, n=inflight, device=dev,
                          wait_for_caller=bool(int(os.environ.get("BFA_BENCH_WAIT_CALLER", "0"))))
    aus = bif.decoders
    au = aus[0]
    lib = _lib.lib()
    hs = [_lib.handle(rk.dev.index, k) for k in range(len(aus))]
    if os.environ.get("BFA_BENCH_PRECREATE"):  # (A/B: the handles' streams created up front, as until round 5)
        for k in range(len(aus)):
            _lib.check(lib.bfa_set_option(hs[k], _lib.OPT_PRECREATE_STREAMS, int(os.environ["BFA_BENCH_PRECREATE"])), hs[k], "bfa_set_option")
    if os.environ.get("BFA_BENCH_ROUTING"):  # (A/B: BFA_OPT_WINDOW_ROUTING of every handle in flight)
        for k in range(len(aus)):
            _lib.set_window_routing(rk.dev.index, k, int(os.environ["BFA_BENCH_ROUTING"]))
    # the lengths are known on the host (constant here), and the synthetic targets never contain SIL:
    # tell the library which K1 register class occurs so that it does not launch the empty ones
    hint = au.viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=(None if args.no_window else C))
    if args.no_uniform_hint:  # A/B: workgroup w = utterance w instead of one contiguous eighth of the batch per XCD
        hint &= ~_lib.HINT_UNIFORM_LENGTHS
    counter = [0]
    last = [None]

    # The reference never aligns without scoring (core.py:902-937: decode_alignments, then _calculate_confidences of the
    # aligned tuples): a step is K1 (forward pass) + K2 (walk, tuples) + K3 (confidence pass) on the batch's stream, and
    # `value` counts aligned AND scored frames.  --no-confidences times K1 + K2 only; the default run reports that figure
    # too (`alignment_only`, a second set of windows).
    with_conf = [not args.no_confidences]

    def step(i):
        lp, tk = bufs[i % nbuf]
        if inflight <= 1:
            r = au.decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint)
            if with_conf[0]:
                r.conf, r.conf_status = calculate_confidences_batch(lp, r.segs, r.seg_count, T_rows=T_len)
            return r
        return bif.submit(lp, tk, T_len, S_len, confidences=with_conf[0], class_mask=hint)

    def run_steps(n):
        for _ in range(n):
            last[0] = step(counter[0])
            counter[0] += 1

    def check_status(r):
        if r is not None and not os.environ.get("BFA_HIP_LIBRARY"):  # (kernel-time experiments with a stubbed role produce garbage)
            st = r.status.cpu().numpy()
            assert (st == 0).all(), f"alignment failed on the bench workload: {np.unique(st)}"

    # ---- W untimed warm-up steps
    run_steps(args.warmup)
    torch.cuda.synchronize()
    check_status(last[0])
    # ---- window 0: the K steps right behind the W warm-up steps, i.e. the first milliseconds of load.  The power management
    # reacts to the load step: K1 starts at its steady duration, rises by ~15 % between ~2 and ~15 ms and settles after
    # ~30-50 ms (profiles/r02_k1_series.txt).  It is REPORTED (`first_window`) but is not `value`: the steps then keep
    # running, untimed, for --settle-ms; everything that ran before the reported windows is counted in
    # `warmup_effective_steps`.
    first_t, _ = _timed_windows(rk, run_steps, K, 1)
    settle_steps = 0
    if args.settle_ms > 0:
        s0 = time.perf_counter()
        while (time.perf_counter() - s0) * 1e3 < args.settle_ms:
            run_steps(8)
            settle_steps += 8
            torch.cuda.synchronize()
    warm_eff = counter[0]
    # ---- the reported windows: ceil(min_timed_steps / K) windows of EXACTLY K steps each (>= 100
    # back-to-back iterations), with NO instrumentation inside them: the HIP event pair around K1 that the profile leg
    # uses costs the headline step 1-2 % and a 256-utterance step 9 % (0.085 against 0.077 ms; BFA_BENCH_K1_EVERY=1 brings
    # it back for an A/B)
    n_windows = max(1, -(-args.min_timed_steps // K))
    k1_every = int(os.environ.get("BFA_BENCH_K1_EVERY", "0"))
    for h in hs:
        lib.bfa_profile_enable(h, k1_every)
    torch.cuda.synchronize()
    win_t, issue_t = _timed_windows(rk, run_steps, K, n_windows)
    check_status(last[0])
    res = last[0]
    if with_conf[0]:
        assert int((res.conf_status != 0).sum().item()) == 0, "the confidence pass reported an error on the bench workload"
    # ---- the same windows without the confidence pass (K1 + K2 only: what rounds 1-3 reported as `value`)
    only_ms = None
    if with_conf[0]:
        with_conf[0] = False
        run_steps(max(4, inflight))
        torch.cuda.synchronize()
        only_t, _ = _timed_windows(rk, run_steps, K, n_windows)
        only_ms = float(np.sum(only_t)) / (K * n_windows) * 1e3
        with_conf[0] = True
    # ---- one more window of K steps, not part of `value`, with K1 of every step bracketed by HIP events on its launch
    # stream: how long a launch shares the machine with the others in flight, and for how long any K1 is running
    for h in hs:
        lib.bfa_profile_collect(h, (ctypes.c_float * 8)(), 8)  # (drops whatever an A/B run recorded above)
        lib.bfa_profile_enable(h, 1)
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    run_steps(K)
    torch.cuda.synchronize()
    spans = []
    cap = K + 8
    for h in hs:
        lib.bfa_profile_enable(h, 0)
        a0 = (ctypes.c_float * cap)()
        a1 = (ctypes.c_float * cap)()
        n = lib.bfa_profile_collect_spans(h, ctypes.c_void_p(base.cuda_event), a0, a1, cap)
        spans += [(float(a0[i]), float(a1[i])) for i in range(n)]
    total_steps = K * n_windows
    elapsed = float(np.sum(win_t))
    win_ms = [t / K * 1e3 for t in win_t]
    k1_inflight = [b - a for a, b in spans]
    busy_ms = _union_ms(spans)
    if os.environ.get("BFA_BENCH_DUMP_K1"):
        print("k1 series (ms), reported windows:", " ".join(f"{v:.3f}" for v in k1_inflight), file=sys.stderr)

    # ---- the kernel leg: the SAME steps with ONE batch in flight (plain stream-ordered calls on the caller's stream), K1
    # bracketed by HIP events on that stream.  Nothing overlaps a K1 launch here, so its mean duration is what
    # `rocprofv3 --kernel-trace --stats -- python bench.py --inflight 1` reports for the kernel
    # (profiles/r03_headline_inflight1_kernel_stats.csv), and roofline.achieved = algorithmic bytes / that mean.
    n_leg = max(20, min(args.kernel_leg_steps, 400))
    torch.cuda.synchronize()
    for i in range(4):
        au.decode_alignments_device(*bufs[i % nbuf], T_len, S_len, class_mask=hint)
    torch.cuda.synchronize()
    lib.bfa_profile_enable(hs[0], 1)
    l0 = time.perf_counter()
    for i in range(n_leg):
        au.decode_alignments_device(*bufs[i % nbuf], T_len, S_len, class_mask=hint)
    torch.cuda.synchronize()
    leg_ms = (time.perf_counter() - l0) / n_leg * 1e3
    lib.bfa_profile_enable(hs[0], 0)
    kbuf = (ctypes.c_float * n_leg)()
    nk = lib.bfa_profile_collect(hs[0], kbuf, n_leg)
    k1_alone = [float(kbuf[i]) for i in range(nk)]
    kernel_ms = float(np.mean(k1_alone)) if nk else float("nan")
    if os.environ.get("BFA_BENCH_DUMP_K1"):
        print(f"k1 series (ms), kernel leg, buffers cycle mod {nbuf}:", " ".join(f"{v:.3f}" for v in k1_alone), file=sys.stderr)
        print("k1 mean per buffer (ms):", " ".join(f"{np.mean(k1_alone[j::nbuf]):.4f}" for j in range(nbuf)), file=sys.stderr)

    # confidence pass (utils._calculate_confidences), timed separately: it is a separate reference call
    lp0, _ = bufs[(counter[0] - 1) % nbuf]
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(5):
        conf, _st = calculate_confidences_batch(lp0, res.segs, res.seg_count)
    torch.cuda.synchronize()
    conf_ms = (time.perf_counter() - c0) / 5 * 1e3

    # final gather of the (small) result records over RCCL, outside the timed steps
    gather_ms = None
    if dist is not None:
        # every rank holds B utterances of <= S tuples: the record bounds need no exchange
        gidx = torch.arange(rank * B, (rank + 1) * B, dtype=torch.int32, device=dev)
        gkw = dict(n_cap=B, tuple_cap=B * res.segs.shape[1])
        gather_results(res.segs, res.seg_count, conf, gidx, world * B, **gkw)  # warm (communicator set-up)
        torch.cuda.synchronize()
        rk.barrier()
        g0 = time.perf_counter()
        out = gather_results(res.segs, res.seg_count, conf, gidx, world * B, **gkw)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            gs, gc, _gf = out.to_padded(res.segs.shape[1])
            gs, gc = gs.to(dev), gc.to(dev)   # (host records under --oversubscribe)
            k = torch.arange(res.segs.shape[1], device=dev)[None, :] < res.seg_count[:, None]
            assert gs.shape[0] == world * B and torch.equal(gs[:B][k], res.segs[k]) \
                and torch.equal(gc[:B], res.seg_count), "gathered records differ from rank 0's own results"

    frames_per_step = B * T
    L = 4 * S + 1
    bytes_per_frame = 4 * C + (L + 3) // 4 + 8
    alg_bytes = frames_per_step * bytes_per_frame
    # the confidence pass reads ONE float per frame a tuple covers -- a 64-byte DRAM sector each -- and writes a float per tuple
    covered = int((res.segs[:, :, 2] - res.segs[:, :, 1]).clamp(min=0).mul(
        torch.arange(res.segs.shape[1], device=dev).unsqueeze(0) < res.seg_count.unsqueeze(1)).sum().item())
    conf_bytes = 64 * covered + 4 * int(res.seg_count.sum().item()) if not args.no_confidences else 0
    value = world * frames_per_step * total_steps / elapsed
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if nk else None
    busy_per_launch = busy_ms / len(spans) if spans else None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:  # timed at N=1 only
        cpu = cpu_baseline_leg(args, bufs, lambda i: au.decode_alignments_device(*bufs[i % nbuf], T_len, S_len, class_mask=hint),
                               B, T, S, C, blank, sil)

    # HBM bytes of one K1 launch from the PMC passes of tools/profile.sh (committed under profiles/), if it was
    # taken on this workload; FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes
    traffic, tfile = None, None
    for name in ("r06_k1_traffic.json", "r05_k1_traffic.json", "r04_k1_traffic.json", "r03_k1_traffic.json", "r02_k1_traffic.json", "r01_k1_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath) and (B, T, S, C) == (4096, 1000, 40, 67) and not args.row_pitch:
            traffic, tfile = json.load(open(tpath))["traffic_bytes_per_launch"], name
            break

    copy_peak = measured_copy_peak(dev, lib, hs[0]) if rank == 0 else None
    valu_insts, valu_file = k1_valu_from_pmc() if (B, T, S, C) == (4096, 1000, 40, 67) else (None, None)
    valu = None
    if valu_insts and nk:
        floor_ms = valu_insts / (N_SIMD * SIMD_CLOCK_HZ / 4) * 1e3
        valu = {"insts_per_launch": valu_insts, "source": f"SQ_INSTS_VALU, rocprofv3 PMC, profiles/{valu_file}",
                "insts_per_frame": valu_insts / frames_per_step, "issue_floor_ms": floor_ms,
                "issue_floor_what": f"insts / ({N_SIMD} SIMDs x {SIMD_CLOCK_HZ / 1e9:g} GHz / 4 cycles per wave64 instruction)",
                "kernel_over_floor": kernel_ms / floor_ms}

    soft = softness_record(args, au.viterbi_decoder, bufs[(counter[0] - 1) % nbuf], T_len, S_len, res) if rank == 0 else None
    ranks = rk.describe()
    if rank == 0:
        kname = "k_dp4w<2,4,3> (K1 banded Viterbi forward, sliding-window consumer)" if (B, T, S, C) == (4096, 1000, 40, 67) \
            else "K1 banded Viterbi forward (class kernel of this shape)"
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors", "value": value,
            "unit": "aligned frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "timed_steps_total": total_steps,
            "timed_steps_what": f"{n_windows} windows of exactly --steps {K} steps are timed; ms_per_step = their total time / {total_steps}",
            "ms_per_step": elapsed / total_steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch={B} T={T} |tokens|={S} ph66 (C={C}) per GPU, reference-default flags "
                                   f"(boost+floor+truly_forced, anchors=10, no SIL in targets -> standard mode)",
                       "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, no data-path collective",
                       "batches_in_flight": inflight, "row_pitch_floats": args.row_pitch or C,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")},
            "timing": {"what": f"{n_windows} windows of exactly {K} steps, each bracketed by barrier + synchronize on both "
                               f"sides (max over ranks); value = frames of all windows / sum of the window times",
                       "windows": n_windows, "timed_steps_total": total_steps,
                       "window_ms_per_step": {"mean": float(np.mean(win_ms)), "min": float(np.min(win_ms)),
                                              "max": float(np.max(win_ms)), "all": win_ms},
                       "warmup_requested_steps": args.warmup, "warmup_effective_steps": warm_eff,
                       "warmup_effective_what": f"{args.warmup} warm-up steps + the first window ({K} steps, reported below) + "
                                                f"{settle_steps} untimed steps over --settle-ms {args.settle_ms:g}",
                       "first_window": {"what": f"the {K} steps right after the {args.warmup} warm-up steps (power-management "
                                                "transient of the first ~30 ms of load)",
                                        "ms_per_step": first_t[0] / K * 1e3,
                                        "value": world * frames_per_step * K / first_t[0]},
                       "host_issue_ms_per_step": float(np.sum(issue_t)) / total_steps * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "measured_copy_peak": copy_peak["gbs_best"] if copy_peak else None,
                         "measured_copy_peak_what": "float4 copy kernel over 1 GiB in this process (bfa_profile_copy), read + "
                                                    "written bytes / best of 12 launches; the data-sheet peak is `peak`",
                         "measured_copy": copy_peak,
                         "frac_of_measured": (achieved / copy_peak["gbs_best"]) if (achieved and copy_peak) else None,
                         "valu": valu,
                         "traffic_unit": f"bytes per K1 launch (rocprofv3 PMC, profiles/{tfile})" if tfile else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_frame": bytes_per_frame,
                         "kernel": kname,
                         "what": "achieved = algorithmic bytes of a K1 launch / mean K1 launch duration in the kernel leg: "
                                 f"{nk} stream-ordered steps with ONE batch in flight inside this run, K1 bracketed by HIP "
                                 "events on its launch stream -- no launch overlaps another, so this is the figure "
                                 "`rocprofv3 --kernel-trace --stats -- python bench.py --inflight 1` reports",
                         "kernel_ms": kernel_ms, "kernel_ms_stats": _stats(k1_alone), "kernel_ms_samples": int(nk),
                         "kernel_ms_per_buffer": [float(np.mean(k1_alone[j::nbuf])) for j in range(nbuf)] if nk else None,
                         "kernel_ms_per_buffer_what": "the kernel leg cycles through the resident batches; K1 takes 0.32 or 0.34 ms depending on where a batch physically lives (profiles/r03_placement.txt)",
                         "kernel_leg_ms_per_step": leg_ms,
                         "whole_step_frac": (alg_bytes + conf_bytes) / (elapsed / total_steps) / 1e9 / HBM_PEAK_GBS,
                         "whole_step_what": "K1 + K2 + K3 of a step (" + ("with" if not args.no_confidences else "without") +
                                            " the confidence pass): algorithmic bytes of K1 plus one 64-byte sector per frame a "
                                            "tuple covers / ms_per_step",
                         "confidence_pass_algorithmic_bytes": conf_bytes,
                         "in_flight": {"what": f"K1 brackets of one more window of {K} steps run after the reported ones (the "
                                               "reported windows carry no instrumentation): with several batches in flight "
                                               "the launches overlap and share the machine (duration > busy time per launch)",
                                       "launch_ms_stats": _stats(k1_inflight),
                                       "busy_ms_per_launch": busy_per_launch,
                                       "launches_running_on_average": (sum(k1_inflight) / busy_ms) if busy_ms > 0 else None,
                                       "frac_by_busy_time": (alg_bytes / (busy_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                       if busy_per_launch else None}},
            "cpu_baseline": cpu,
            "softness": soft,
            "reference_cpu_baseline": _reference_cpu_record(),
            "step": "K1 + K2 + K3: alignment, walk / tuples, confidence pass (core.py:902-937)" if not args.no_confidences
                    else "K1 + K2: alignment, walk / tuples (--no-confidences)",
            "alignment_only": ({"what": "the same windows with K1 + K2 only (no confidence pass): the `value` of rounds 1-3",
                                "ms_per_step": only_ms, "value": world * frames_per_step / (only_ms * 1e-3)} if only_ms else None),
            "confidence_pass_ms": conf_ms,
            "gather_ms": gather_ms,
            "rccl_ranks_seen": rk.group_record()["rccl_ranks_seen"], "group": rk.group_record(),
            "ranks": ranks,
        }
        print(json.dumps(line))


def softness_record(args, vd, buf, T_len, S_len, res, n=64):
    """How sharp the synthetic posteriors of this run are and what that made the library do: the
    generator's peak / sigma, the mean log-probability per frame of the aligned path on the prepared emissions, the share of a
    sample whose path ends at the reference's -1000 sentinel (forced_alignment.py:23,656-682), and the window counters of the
    last call (bfa_call_counters)."""
    try:
        lp, tk = buf
        n = min(n, lp.shape[0])
        m = vd.prepare_emissions(lp[:n], tk[:n], T_len[:n], S_len[:n])
        fph = res.frame_phonemes[:n].long().clamp(min=0)
        g = m.gather(2, fph.unsqueeze(-1)).squeeze(-1).double()
        mask = torch.arange(lp.shape[1], device=lp.device)[None, :] < T_len[:n, None]
        tot = (g * mask).sum(1)
        cnt = res.call_counters()
        return {"peak": args.peak, "sigma": args.sigma, "what": "logits = N(0, sigma) + peak * onehot(planted path)",
                "path_logp_per_frame": float(tot.sum() / mask.sum()), "sample_share_at_sentinel": float((tot <= -1000.0).double().mean()),
                "last_call": {k: cnt[k] for k in ("items", "redone_full", "redone_exact", "exact_done", "exact_alive")},
                "last_call_what": "redone_*: fast windows that gave up (dead / doomed) and were aligned again; exact_done / exact_alive: "
                                  "items the exact window aligned (reruns + calls routed there at once, BFA_OPT_WINDOW_ROUTING) / of "
                                  "those above the sentinel"}
    except Exception as e:  # (a measurement aid, never the reason a bench line is missing)
        return {"peak": args.peak, "sigma": args.sigma, "error": repr(e)}


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


def c4_plan(n_total, world, seed, chunk, halves=1):
    """Host-side plan, identical on every rank: lengths, the LPT partition (sharding.shard_utterances on the DP work
    T*(4S+1); a rank's predicted time max(chain, work / machine rate) is reported beside the measured one), and per shard
    `halves` sub-shards that are
    aligned side by side (every halves-th utterance of the length-sorted shard: the same length mix in each), each cut
    into the length-sorted sub-batches ("chunks") one bfa_align_batch call takes.  plans[rank][half] = list of chunks."""
    from bournemouth_forced_aligner_amd.sharding import shard_utterances, utterance_cost
    T, S = c4_lengths(n_total, seed)
    shards = shard_utterances(T, S, world)
    plans = []
    for s in shards:
        order = s[np.argsort(-T[s], kind="stable")]  # longest first: similar lengths share a call (less padding,
        subs = []                                    # fewer K1 classes per call)
        for hf in range(halves):
            o = order[hf::halves]
            subs.append([o[i:i + chunk] for i in range(0, len(o), chunk)])
        plans.append(subs)
    cost = utterance_cost(T, S)
    loads = np.array([cost[s].sum() for s in shards], np.float64)
    return T, S, shards, plans, loads


def c4_record_bounds(shards, S, cap, extra=0):
    """(n_cap, tuple_cap) of the packed result records, the same on every rank WITHOUT an exchange: every rank knows the
    partition, an utterance of S tokens yields at most min(S, cap) tuples (ignore_noise: blank runs are not emitted,
    forced_alignment.py:822-831) plus `extra` rows the bench lets ride along."""
    n_cap = max(len(s) for s in shards)
    tuple_cap = max(int((np.minimum(S[s], cap) + extra).sum()) for s in shards)
    return max(n_cap, 1), max(tuple_cap, 1)


def c4_main(args, rk):
    dev, rank, world, dist = rk.dev, rk.rank, rk.world, rk.dist
    C, seed, n_total = args.classes, args.seed, args.global_batch
    blank = C - 1
    halves = max(1, args.halves)
    T, S, shards, plans, loads = c4_plan(n_total, world, seed, args.chunk, halves)
    mine = shards[rank]
    # every rank must have computed the same partition
    digest = _shard_digest(np.concatenate([np.asarray([len(x) for x in shards], np.int64)] + list(shards)))
    _, digests = rk.max_over_ranks(float(digest % (1 << 52)))
    shard_agree = len(set(digests)) == 1
    assert shard_agree, f"ranks disagree on the shard partition: {digests}"
    allidx = np.sort(np.concatenate(shards))
    assert np.array_equal(allidx, np.arange(n_total)), "the shards are not a partition of the batch"

    from bournemouth_forced_aligner_amd.sharding import gather_results, predict_rank_ms
    cap = int(S.max()) + 2
    if rk.dry:
        # no GPU work: exercise launch, partition agreement and the gather plumbing with fabricated records
        segs = torch.zeros((len(mine), cap, 4), dtype=torch.int32)
        segs[:, 0, 0] = torch.from_numpy(mine).to(torch.int32)
        cnt = torch.from_numpy(np.minimum(S[mine], cap)).to(torch.int32)
        n_cap, tuple_cap = c4_record_bounds(shards, S, cap, extra=0)
        out = gather_results(segs, cnt, None, torch.from_numpy(mine), n_total, n_cap=n_cap, tuple_cap=tuple_cap)
        ok = None
        if rank == 0:
            gs, gc, _ = out.to_padded(cap)
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
    vd.window_max_frames = args.win_frames or None   # (A/B: how long an utterance may be for K1 to try its sliding window)
    vd.window_max_tokens = args.win_tokens or None
    lib = _lib.lib()
    h = _lib.handle(rk.dev.index)

    # ---- synthesis: each rank makes only its own utterances (sub-batches of <= 512 to bound the temporaries)
    t_s0 = time.perf_counter()
    chunks = []
    for hf, ch in [(hf, ch) for hf in range(halves) for ch in plans[rank][hf]]:
        if len(ch) == 0:
            continue
        Tp, Sp = int(T[ch].max()), int(S[ch].max())
        lps, tks = [], []
        for i in range(0, len(ch), 512):
            sub = ch[i:i + 512]
            lp, tk = c4_utterances(sub, T[sub], S[sub], C, seed, dev, Tpad=Tp, Spad=Sp, peak=args.peak, sigma=args.sigma)
            lps.append(lp)
            tks.append(tk)
        lp = torch.cat(lps, 0) if len(lps) > 1 else lps[0]
        tk = torch.cat(tks, 0) if len(tks) > 1 else tks[0]
        del lps, tks
        Td = torch.from_numpy(T[ch].astype(np.int32)).to(dev)
        Sd = torch.from_numpy(S[ch].astype(np.int32)).to(dev)
        hint = vd.class_mask_hint(T[ch], S[ch], has_sil=False, n_classes=C)
        chunks.append(dict(idx=ch, lp=lp, tk=tk, Td=Td, Sd=Sd, hint=hint, csum=input_checksum(lp, T[ch]), half=hf))
    torch.cuda.synchronize()
    synth_s = time.perf_counter() - t_s0
    my_frames = int(T[mine].sum())
    my_bytes = int(((4 * C + (4 * S[mine] + 1 + 3) // 4 + 8) * T[mine]).sum())

    # --inflight k: k steps in flight, each on its own stream with its own decoder (workspace) and library handle (aux
    # streams).  A rank's shard of a sharded batch is bound by the chain of its longest utterance, not by the machine;
    # a service that aligns a stream of such batches overlaps them.  `value` stays frames / wall time.
    # --halves h (default 2): the rank's shard as h sub-shards side by side, each with its own stream, decoder and library
    # handle: the chains of one sub-shard's long utterances run beside the short work of the other.
    nfl = max(1, args.inflight or 1)
    nlanes = nfl * halves
    aus = [au]
    for k in range(1, nlanes):
        a2 = AlignmentUtils(blank_id=blank, silence_id=0)
        a2.viterbi_decoder.handle_slot = k
        a2.viterbi_decoder.window_max_frames, a2.viterbi_decoder.window_max_tokens = vd.window_max_frames, vd.window_max_tokens
        aus.append(a2)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nlanes)] if nlanes > 1 else None
    step_no = [0]

    def step():
        k = step_no[0] % nfl
        step_no[0] += 1
        if streams is None:
            return [au.decode_alignments_device(c["lp"], c["tk"], c["Td"], c["Sd"], class_mask=c["hint"], seg_cap=cap)
                    for c in chunks]
        out = []
        for c in chunks:
            lane = k * halves + c["half"]
            with torch.cuda.stream(streams[lane]):
                out.append(aus[lane].decode_alignments_device(c["lp"], c["tk"], c["Td"], c["Sd"], class_mask=c["hint"],
                                                              seg_cap=cap))
        return out

    for _ in range(max(1, args.warmup)):
        res = step()
    torch.cuda.synchronize()
    for r in res:
        st = r.status.cpu().numpy()
        assert (st == 0).all(), f"alignment failed on the C4 workload: {np.unique(st)}"

    rk.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):  # (no instrumentation inside the timed steps)
        res = step()
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    rk.barrier()
    elapsed = time.perf_counter() - t0
    # the K1 time of a step (first class kernel's start to the last one's end, HIP events): two more steps, not part of `value`
    nprof = 2
    lib.bfa_profile_enable(h, 1)
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    lib.bfa_profile_enable(h, 0)
    ncall = nprof * len(chunks)
    k1 = (ctypes.c_float * max(1, ncall))()
    nk1 = lib.bfa_profile_collect(h, k1, ncall)
    k1_step_ms = float(np.sum([k1[i] for i in range(nk1)])) / nprof if nk1 else float("nan")
    elapsed, _ = rk.max_over_ranks(elapsed)
    _, rank_ms = rk.max_over_ranks(mine_s / args.steps * 1e3)

    # ---- final gather of the result records (outside the steps; timed on its own, after a warm-up gather that
    # pays for the communicator set-up)
    segs = torch.cat([r.segs for r in res], 0) if len(res) > 1 else res[0].segs
    cnt = torch.cat([r.seg_count for r in res], 0) if len(res) > 1 else res[0].seg_count
    gidx = torch.from_numpy(np.concatenate([c["idx"] for c in chunks]).astype(np.int32)).to(dev)
    csum = torch.cat([c["csum"] for c in chunks], 0)
    # the input checksum rides along as one more "tuple" behind each utterance's own (row count[b] < cap is free)
    cs62 = csum & ((1 << 62) - 1)
    rows = torch.arange(segs.shape[0], device=dev)
    segs[rows, cnt.long(), 0] = (cs62 & ((1 << 31) - 1)).to(torch.int32)
    segs[rows, cnt.long(), 1] = (cs62 >> 31).to(torch.int32)
    cnt_x = cnt + 1
    # record bounds every rank derives from the partition it already knows: no size exchange
    n_cap, tuple_cap = c4_record_bounds(shards, S, cap, extra=1)
    gkw = dict(n_cap=n_cap, tuple_cap=tuple_cap)
    gather_results(segs, cnt_x, None, gidx, n_total, **gkw)
    torch.cuda.synchronize()
    gms = []
    for _ in range(5):
        rk.barrier()
        g0 = time.perf_counter()
        out = gather_results(segs, cnt_x, None, gidx, n_total, **gkw)   # one pack kernel + one collective
        torch.cuda.synchronize()
        gms.append((time.perf_counter() - g0) * 1e3)
    gather_ms, _ = rk.max_over_ranks(float(np.median(gms)))
    index_ms = host_ms = payload = None
    if rank == 0:
        g0 = time.perf_counter()
        out.index()                                                        # one kernel: (owner, offset, count) per utterance
        torch.cuda.synchronize()
        index_ms = (time.perf_counter() - g0) * 1e3
        g0 = time.perf_counter()
        out.host()                                                         # the records on the host (pageable memory)
        host_ms = (time.perf_counter() - g0) * 1e3
        payload = int(out.records.numel() * 4)
        assert not out.overflowed(), "a rank's tuples did not fit the agreed record bound"

    total_frames = int(T.sum())
    total_bytes = int(((4 * C + (4 * S + 1 + 3) // 4 + 8) * T).sum())
    step_s = elapsed / args.steps
    parity = None
    if rank == 0:
        parity = c4_parity_sample(args, T, S, out, cap, dev, C, seed)
    ranks = rk.describe()
    if rank == 0:
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors", "value": total_frames / step_s,
            "unit": "aligned frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: global batch={n_total} mixed-length T~U[200,3000] S=T//25 ph66 (C={C}), seed {seed}, "
                                   f"reference-default flags; {nfl} step(s) in flight; LPT-sharded over {world} rank(s), "
                                   f"per rank {halves} sub-shard(s) side by side in {len(chunks)} length-sorted calls of "
                                   f"<= {args.chunk} utterances",
                       "global_batch": n_total, "sub_shards_in_flight": halves,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "parallelism": f"utterance-sharded x{world} (LPT on T*(4S+1)), "
                                      f"no data-path collective, final gather of records"},
            "predicted_rank_ms": [predict_rank_ms(T[s], S[s]) for s in shards],
            "frames_per_step": total_frames,
            "value_with_gather": total_frames / (step_s + gather_ms * 1e-3),
            "gather_ms": gather_ms,
            "gather": {"what": "bfa_pack_results (one kernel) + one torch.distributed.gather of equal-size packed records; "
                               "median of 5, max over ranks; the receiver keeps the records as they arrive",
                       "payload_bytes": payload, "record_bounds": {"n_cap": n_cap, "tuple_cap": tuple_cap},
                       "index_ms": index_ms, "records_to_host_ms": host_ms},
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
            "rccl_ranks_seen": rk.group_record()["rccl_ranks_seen"], "group": rk.group_record(),
            "ranks": ranks,
        }
        print(json.dumps(line))


def c4_parity_sample(args, T, S, out, cap, dev, C, seed):
    """Rank 0, after the gather: a stratified sample (every length stratum + the 32 longest utterances) is synthesised
    again from the global indices, its input checksum compared with the one the owning rank gathered, and the
    gathered records compared with the oracle's."""
    from oracle import oracle as ora
    n_total = len(T)
    n = min(args.parity_sample, n_total)
    order = np.argsort(T, kind="stable")
    longest = order[-min(32, n):]
    strat = order[np.linspace(0, n_total - 1, max(0, n - len(longest))).astype(np.int64)]
    sample = np.unique(np.concatenate([longest, strat])) if n < n_total else np.arange(n_total)  # (--parity-sample >= batch: everything)
    prm = ora.make_params(C - 1, 0)
    mism, bad_inputs, frames = 0, 0, 0
    w = 0.0
    for i in range(0, len(sample), 64):
        sub = sample[i:i + 64]
        lp, tk = c4_utterances(sub, T[sub], S[sub], C, seed, dev, peak=args.peak, sigma=args.sigma)
        cs = input_checksum(lp, T[sub]).cpu().numpy()
        lp_h, tk_h = lp.cpu().numpy(), tk.cpu().numpy()
        w0 = time.perf_counter()
        exp = ora.decode_alignments(lp_h, tk_h, T[sub], S[sub], prm, seg_cap=cap)
        w += time.perf_counter() - w0
        for k, g in enumerate(sub):
            frames += int(T[g])
            rows, _ = out.rows(int(g))   # the utterance's tuples + the checksum row, from the packed records
            got_cs = int(rows[-1, 0]) + (int(rows[-1, 1]) << 31)
            if got_cs != int(cs[k]) & ((1 << 62) - 1):
                bad_inputs += 1
            c = int(exp["seg_count"][k])
            if rows.shape[0] - 1 != c or not (rows[:c] == exp["seg"][k, :c]).all():
                mism += 1
    return {"utterances": int(len(sample)), "frames": frames, "longest_T": int(T[sample].max()),
            "mismatching_utterances": mism, "regenerated_inputs_differing": bad_inputs,
            "oracle_s": w, "oracle_frames_per_s_1core": frames / w if w > 0 else None}


# --------------------------------------------------------------------------------------------- realtext
def realtext_main(args, rk):
    """`--config realtext`: what the reference does with a REAL transcript (core.py:897-937): raw logits of both heads
    (ph66 C = 67, groups C = 17), targets with SIL (punctuation) -> the silence-anchored segmented mode
    (forced_alignment.py:268-469) in both heads, then coverage + soft boundaries (core.py:925-931) and confidences
    (core.py:936-937) per head.  One step = ONE bfa_align_heads call with the post-DP stages of each head enqueued behind
    its alignment (bfa_postprocess / bfa_confidences on the head's stream) on device-resident logits; nothing synchronises
    inside a step.  Parity: the first `--parity-sample` utterances of the batch through
    the oracle's whole chain (log_softmax -> decode -> coverage -> soft boundaries -> confidences), both heads."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, calculate_confidences_batch, _lib
    from bournemouth_forced_aligner_amd.forced_alignment import align_heads
    from bournemouth_forced_aligner_amd.utils import postprocess_batch
    from tools.synth import synth_realtext
    dev, rank, world = rk.dev, rk.rank, rk.world
    B, T, S = args.batch, args.frames, args.tokens
    K = args.steps
    nfl = max(1, args.inflight if args.inflight is not None else 3)
    nbuf = max(2, nfl)
    bufs = [synth_realtext(B, T, S, 2003 + 17 * rank + 1000 * i, dev, peak=args.peak, gpeak=max(1.0, args.peak - 2.0), sigma=args.sigma)
            for i in range(nbuf)]
    T_len = torch.full((B,), T, dtype=torch.int32, device=dev)
    S_len = torch.full((B,), S, dtype=torch.int32, device=dev)
    slots = []
    for k in range(nfl):
        ap, ag = AlignmentUtils(blank_id=66, silence_id=0), AlignmentUtils(blank_id=16, silence_id=0)
        ap.viterbi_decoder.handle_slot = ag.viterbi_decoder.handle_slot = k
        slots.append((ap, ag))
        # one call at a time: the heads of a call on two library streams; several in flight: head 1 on the side stream (bfa.h)
        _lib.set_calls_in_flight(rk.dev.index, k, nfl > 1)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)] if nfl > 1 else [None]
    vd = slots[0][0].viterbi_decoder
    hints = [vd.class_mask_hint([T] * B, [S] * B, has_sil=True, n_classes=67),
             vd.class_mask_hint([T] * B, [S] * B, has_sil=True, n_classes=17)]
    soft = 3  # PhonemeTimestampAligner(boundary_softness=3) (core.py:41)
    counter = [0]
    last = [None]

    def one(ap, ag, buf, T_len=T_len, S_len=S_len, hints=hints):
        xp, xg, tp, tg = buf
        if args.separate_post:  # A/B: the post-DP stages as separate calls behind the joined alignment (round 2's order)
            (rp, sp), (rg, sg) = align_heads([ap, ag], [xp, xg], [tp, tg], T_len, S_len, class_masks=hints)
            out = []
            for x, r, st in ((xp, rp, sp), (xg, rg, sg)):
                postprocess_batch(x, S_len, r.segs, r.seg_count, extend=True, boundary_softness=soft, row_stats=st)
                cf, cs = calculate_confidences_batch(x, r.segs, r.seg_count, row_stats=st)
                out.append((r, cf, cs))
            return out
        (rp, _sp), (rg, _sg) = align_heads([ap, ag], [xp, xg], [tp, tg], T_len, S_len, class_masks=hints,
                                           post={"extend": True, "boundary_softness": soft})
        return [(rp, rp.conf, rp.conf_status), (rg, rg.conf, rg.conf_status)]

    nch = max(1, args.chunks)
    cb = -(-B // nch)
    ch_hints = [vd.class_mask_hint([T] * cb, [S] * cb, has_sil=True, n_classes=67),
                vd.class_mask_hint([T] * cb, [S] * cb, has_sil=True, n_classes=17)]
    cc = [0]

    def run_steps(n):
        if nch > 1:  # A/B: the batch as `--chunks` calls of B / chunks utterances, round-robin over the streams in flight
            for _ in range(n):
                i = counter[0]
                counter[0] += 1
                for j in range(nch):
                    k = cc[0] % nfl
                    cc[0] += 1
                    sl = slice(j * cb, min(B, (j + 1) * cb))
                    sub = tuple(t[sl] for t in bufs[i % nbuf])
                    if streams[k] is None:
                        last[0] = one(*slots[k], sub, T_len[sl], S_len[sl], ch_hints)
                    else:
                        with torch.cuda.stream(streams[k]):
                            last[0] = one(*slots[k], sub, T_len[sl], S_len[sl], ch_hints)
            return
        for _ in range(n):
            i = counter[0]
            counter[0] += 1
            k = i % nfl
            if streams[k] is None:
                last[0] = one(*slots[k], bufs[i % nbuf])
            else:
                with torch.cuda.stream(streams[k]):
                    last[0] = one(*slots[k], bufs[i % nbuf])

    run_steps(max(1, args.warmup))
    torch.cuda.synchronize()
    for r, _cf, cs in last[0]:
        st = r.status.cpu().numpy()
        assert (st == 0).all() and int((cs.cpu() != 0).sum()) == 0, f"alignment failed on the realtext workload: {np.unique(st)}"
    first_t, _ = _timed_windows(rk, run_steps, K, 1)
    settle_steps = 0
    if args.settle_ms > 0:
        s0 = time.perf_counter()
        while (time.perf_counter() - s0) * 1e3 < args.settle_ms:
            run_steps(4)
            settle_steps += 4
            torch.cuda.synchronize()
    warm_eff = counter[0]
    n_windows = max(1, -(-args.min_timed_steps // K))
    win_t, issue_t = _timed_windows(rk, run_steps, K, n_windows)
    total_steps = K * n_windows
    elapsed = float(np.sum(win_t))
    win_ms = [t / K * 1e3 for t in win_t]
    modes = [r.mode.cpu().numpy() for r, _cf, _cs in last[0]]

    # algorithmic bytes per frame (SURVEY.md 8(d) + the "-sil" prepass): each head's matrix read once by K1, the
    # phoneme head's once more by the P(SIL) prepass K0, 2-bit backpointers and the two framewise outputs per head
    L = 4 * S + 1
    bpf = 4 * 67 + 4 * 17 + 4 * 67 + 2 * ((L + 3) // 4) + 2 * 8
    frames = B * T
    step_s = elapsed / total_steps
    # ---- parity: one more step on buffer 0, the first n utterances against the oracle's whole chain
    parity = None
    if rank == 0 and args.parity_sample > 0:
        parity = realtext_parity(args, slots[0], one, bufs[0], T, S, soft)
    ranks = rk.describe()
    if rank == 0:
        line = {
            "metric": "aligned frames/sec (whole node) on ph66 posteriors, real-text workload (both heads from raw logits, "
                      "SIL in the targets)", "value": world * frames / step_s,
            "unit": "aligned frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"realtext: batch={B} T={T} |tokens|={S}, ph66 head (C=67) + group head (C=17) from RAW "
                                   f"logits, SIL at ~1/12 of the target positions with planted 12-40-frame silences, "
                                   f"reference-default flags; step = bfa_align_heads + bfa_postprocess x2 + "
                                   f"bfa_confidences x2; {nfl} step(s) in flight",
                       "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, no data-path collective",
                       "steps_in_flight": nfl, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")},
            "timing": {"windows": n_windows, "timed_steps_total": total_steps,
                       "window_ms_per_step": {"mean": float(np.mean(win_ms)), "min": float(np.min(win_ms)),
                                              "max": float(np.max(win_ms)), "all": win_ms},
                       "warmup_requested_steps": args.warmup, "warmup_effective_steps": warm_eff,
                       "first_window_ms_per_step": first_t[0] / K * 1e3,
                       "host_issue_ms_per_step": float(np.sum(issue_t)) / total_steps * 1e3},
            "roofline": {"bound": "hbm", "achieved": frames * bpf / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frames * bpf / step_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step (K0 row pass + planner + piece DPs + walks of both heads, post-DP stages)",
                         "algorithmic_bytes_per_frame": bpf,
                         "algorithmic_bytes_what": "4*67 (K1, ph66) + 4*17 (K1, groups) + 4*67 (K0 prepass, ph66) + "
                                                   "2*ceil(L/4) backpointers + 2*8 framewise outputs"},
            "segmented_utterances": {"ph66": int((modes[0] == 1).sum()), "groups": int((modes[1] == 1).sum()), "of": B},
            "parity": parity,
            "ranks": ranks,
        }
        print(json.dumps(line))


def realtext_parity(args, slot, one, buf, T, S, soft):
    from oracle import oracle as ora
    n = min(args.parity_sample, buf[0].shape[0])
    got = one(*slot, buf)
    torch.cuda.synchronize()
    mism = {"ph66": 0, "groups": 0}
    conf_bad = {"ph66": 0, "groups": 0}
    max_conf_diff = 0.0
    tuples = 0
    w0 = time.perf_counter()
    for hi, (name, blank) in enumerate((("ph66", 66), ("groups", 16))):
        x = buf[hi][:n].cpu().numpy()
        tk = buf[2 + hi][:n].cpu().numpy()
        r, cf, _cs = got[hi]
        gs, gc, gcf = r.segs[:n].cpu().numpy(), r.seg_count[:n].cpu().numpy(), cf[:n].cpu().numpy()
        prm = ora.make_params(blank, 0)
        for b in range(n):
            lp = ora.log_softmax_rows(x[b])
            res = ora.decode_alignments(lp[None], tk[b][None], [T], [S], prm)
            rows = ora.segments_as_lists(res)[0]
            cov = ora.ensure_target_coverage_default(rows, S)
            ext = ora.extend_soft_boundaries(lp, cov, soft) if cov else []
            mine = [tuple(int(v) for v in row) for row in gs[b, :gc[b]]]
            if mine != [tuple(int(v) for v in e[:4]) for e in ext]:
                mism[name] += 1
                continue
            rc, c, _s, _e = ora.confidences(lp, ext)
            tuples += len(ext)
            d = float(np.abs(gcf[b, :len(ext)] - c).max()) if len(ext) else 0.0
            max_conf_diff = max(max_conf_diff, d)
            if rc != 0 or d > 1e-4:
                conf_bad[name] += 1
    return {"utterances": n, "heads": 2, "tuples_compared": tuples, "mismatching_utterances": mism,
            "confidence_beyond_1e-4": conf_bad, "max_confidence_abs_diff": max_conf_diff,
            "oracle_s": time.perf_counter() - w0,
            "what": "rows after coverage + soft boundaries bit-exact, confidences within 1e-4, vs oracle/ (log_softmax -> "
                    "decode_alignments -> ensure_target_coverage -> extend_soft_boundaries -> confidences)"}


# ---------------------------------------------------------------------------------------------- c5proxy
def c5proxy_main(args, rk):
    """`--config c5proxy`: the stand-in for BASELINE.json configs[4] (LJSpeech through the real model -- whose checkpoint the
    reference downloads at core.py:260-285 and which is not available offline): what `process_segments` hands the path --
    segments of at most 30 s (T ~ U[300, 1870] frames, README.md:1231), S = T // 12 targets, SIL at the punctuation rate of the
    reference's committed LJSpeech outputs (2 in 110 phonemes) with planted silences, RAW logits of both heads, mixed lengths
    in one call, coverage + soft boundaries + confidences behind each head's alignment (core.py:897-937).  One step = ONE
    bfa_align_heads call over `--batch` utterances per GPU (weak scaling); `--peak` / `--sigma` set the sharpness.  Parity: a
    stratified sample (incl. the longest utterances) through the oracle's whole chain on rank 0."""
    from bournemouth_forced_aligner_amd import AlignmentUtils, _lib
    from bournemouth_forced_aligner_amd.forced_alignment import align_heads
    from tools.synth import synth_realtext_ragged
    from tools.softness import parity_heads, stratified
    dev, rank, world = rk.dev, rk.rank, rk.world
    B, K = args.batch, args.steps
    nfl = max(1, args.inflight if args.inflight is not None else 1)
    xs = synth_realtext_ragged(B, args.tlo, args.thi, args.tok_div, 2004 + 17 * rank, dev, peak=args.peak,
                               gpeak=max(1.0, args.peak - 2.0), sigma=args.sigma)
    bufs, Tl, Sl = xs[:4], xs[4].numpy().astype(np.int64), xs[5].numpy().astype(np.int64)
    xp, xg, tp, tg = bufs
    Td, Sd = xs[4].to(dev), xs[5].to(dev)
    slots, streams = [], []
    for k in range(nfl):
        ap_, ag_ = AlignmentUtils(blank_id=66, silence_id=0), AlignmentUtils(blank_id=16, silence_id=0)
        ap_.viterbi_decoder.handle_slot = ag_.viterbi_decoder.handle_slot = k
        slots.append((ap_, ag_))
        _lib.set_calls_in_flight(rk.dev.index, k, nfl > 1)
        streams.append(torch.cuda.Stream(device=dev) if nfl > 1 else None)
    vd = slots[0][0].viterbi_decoder
    hints = [vd.class_mask_hint(Tl, Sl, has_sil=True, n_classes=67), vd.class_mask_hint(Tl, Sl, has_sil=True, n_classes=17)]
    counter, last = [0], [None]

    def one(k):
        ap_, ag_ = slots[k]
        (rp, _sp), (rg, _sg) = align_heads([ap_, ag_], [xp, xg], [tp, tg], Td, Sd, class_masks=hints,
                                           post={"extend": True, "boundary_softness": 3})
        return rp, rg

    def run_steps(n):
        for _ in range(n):
            k = counter[0] % nfl
            counter[0] += 1
            if streams[k] is None:
                last[0] = one(k)
            else:
                with torch.cuda.stream(streams[k]):
                    last[0] = one(k)

    run_steps(max(1, args.warmup))
    torch.cuda.synchronize()
    for r in last[0]:
        assert bool((r.status.cpu() == 0).all()) and bool((r.conf_status.cpu() == 0).all()), "alignment failed on the c5proxy workload"
    n_windows = max(1, -(-args.min_timed_steps // K))
    win_t, issue_t = _timed_windows(rk, run_steps, K, n_windows)
    total_steps = K * n_windows
    elapsed = float(np.sum(win_t))
    step_s = elapsed / total_steps
    frames = int(Tl.sum())
    nbytes = int(((4 * 67 + 4 * 17 + 4 * 67 + 2 * ((4 * Sl + 1 + 3) // 4) + 2 * 8) * Tl).sum())
    modes = [r.mode.cpu().numpy() for r in last[0]]
    parity = None
    if rank == 0 and args.parity_sample > 0:
        got = one(0)
        torch.cuda.synchronize()
        parity = parity_heads(bufs, Tl, Sl, got, stratified(Tl, max(4, args.parity_sample // 2)))
    ranks = rk.describe()
    if rank == 0:
        print(json.dumps({
            "metric": "aligned frames/sec (whole node) on ph66 posteriors, C5 proxy (mixed-length segments <= 30 s, both heads "
                      "from raw logits, SIL in the targets)", "value": world * frames / step_s, "unit": "aligned frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"c5proxy: batch={B} per GPU, T~U[{args.tlo},{args.thi}], S=T//{args.tok_div}, SIL at 1/40 of the "
                                   f"targets with planted 6-30-frame silences, ph66 head (C=67) + group head (C=17) from RAW logits, "
                                   f"reference-default flags; step = bfa_align_heads with the post-DP stages of both heads; "
                                   f"{nfl} step(s) in flight; logits N(0,{args.sigma:g}) + {args.peak:g} on the planted class",
                       "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, no data-path collective"},
            "timing": {"windows": n_windows, "timed_steps_total": total_steps,
                       "window_ms_per_step": [t / K * 1e3 for t in win_t],
                       "host_issue_ms_per_step": float(np.sum(issue_t)) / total_steps * 1e3},
            "frames_per_step": frames,
            "roofline": {"bound": "hbm", "achieved": nbytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / step_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step (K0 row pass + planner + piece / fallback DPs + walks of both heads, post-DP stages)",
                         "algorithmic_bytes_per_step": nbytes},
            "softness": {"peak": args.peak, "gpeak": max(1.0, args.peak - 2.0), "sigma": args.sigma},
            "segmented_share": {"ph66": float((modes[0] == 1).mean()), "groups": float((modes[1] == 1).mean())},
            "parity": parity, "rccl_ranks_seen": rk.group_record()["rccl_ranks_seen"], "group": rk.group_record(), "ranks": ranks}))


# ----------------------------------------------------------------------------------------------- ragged
def ragged_main(args, rk):
    """Side measurement: throughput on ONE unsorted mixed-length call (prints its own JSON; its parity check against
    the oracle is tests/test_gpu_parity.py::test_mixed_length_shard_parity)."""
    from bournemouth_forced_aligner_amd import AlignmentUtils
    dev, rank = rk.dev, rk.rank
    C, B = args.classes, args.batch
    lp, tk, T_len, S_len = synth_ragged(B, args.tlo, args.thi, C, 1004 + rank, dev, peak=args.peak, sigma=args.sigma, tok_div=args.tok_div)
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
        out = gather_results(segs, cnt, None, gidx, rk.world * B, n_cap=B, tuple_cap=B)
        out = out.to_padded(cap) if rk.rank == 0 else None
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
    ap.add_argument("--config", choices=["headline", "c2", "c4", "realtext", "c5proxy"], default="headline",
                    help="headline = BASELINE configs[2]; c2 = configs[1] (batch 256, T 600, 20 tokens) through the headline "
                         "path; c4 = configs[3]; realtext = both heads from raw logits with SIL in the targets")
    ap.add_argument("--min-timed-steps", type=int, default=100,
                    help="the reported figure covers ceil(this / --steps) windows of exactly --steps steps (SURVEY 8(d): >= 100)")
    ap.add_argument("--kernel-leg-steps", type=int, default=40,
                    help="headline: steps of the one-batch-in-flight leg that prices the kernel for `roofline`")
    ap.add_argument("--hw-queues", type=int, default=None,
                    help="GPU_MAX_HW_QUEUES for this process (the runtime maps HIP streams onto this many hardware queues, "
                         "default 4; read once at HIP initialisation).  Default: the runtime's own, for every config")
    ap.add_argument("--separate-post", action="store_true",
                    help="realtext A/B: bfa_postprocess / bfa_confidences as separate calls after bfa_align_heads")
    ap.add_argument("--row-pitch", type=int, default=0,
                    help="headline A/B: posterior rows padded to this many floats (e.g. 72 = 288-byte rows), 0 = dense")
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
                         "(default: 4 for the headline, 3 for --config realtext, 1 for --config c4)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="utterances timed on the host oracle (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-confidences", action="store_true",
                    help="headline: time K1 + K2 only (by default a step also runs the confidence pass of the aligned tuples)")
    ap.add_argument("--no-window", action="store_true", help="A/B: full state layout instead of the sliding window")
    ap.add_argument("--win-frames", type=int, default=0, help="A/B: window frame limit, 0 = default")
    ap.add_argument("--win-tokens", type=int, default=0, help="A/B: window token limit, 0 = default")
    ap.add_argument("--ragged", action="store_true",
                    help="side measurement: ONE unsorted mixed-length call T~U{200..3000}, S=T//25")
    ap.add_argument("--global-batch", type=int, default=32768, help="c4: utterances over all ranks")
    ap.add_argument("--no-uniform-hint", action="store_true", help="headline A/B: drop BFA_HINT_UNIFORM_LENGTHS from the class hint")
    ap.add_argument("--place-align", type=int, default=0, help="headline A/B: copy the posteriors to addresses aligned to this many bytes (0 = leave them where the allocator put them)")
    ap.add_argument("--place-offset", type=int, default=0, help="headline A/B: ... plus this offset")
    ap.add_argument("--chunks", type=int, default=1, help="realtext A/B: the batch as this many bfa_align_heads calls per step")
    ap.add_argument("--halves", type=int, default=1,
                    help="c4: sub-shards of a rank's shard aligned side by side (own stream / decoder / library handle each)")
    ap.add_argument("--chunk", type=int, default=32768, help="c4: utterances per bfa_align_batch call (one call per rank when the shard is smaller; "
                    "N = 1, same box: 7.14 ms as one call, 7.48 as two of 16384, 7.98 as four of 8192)")
    ap.add_argument("--seed", type=int, default=1004, help="c4: generator seed")
    ap.add_argument("--parity-sample", type=int, default=None,
                    help="utterances rank 0 checks against the oracle (c4: default 256, stratified; realtext: default 512)")
    ap.add_argument("--dry-run", action="store_true", help="launch + partition + gather plumbing on gloo, no GPU work")
    ap.add_argument("--peak", type=float, default=9.0,
                    help="sharpness of the synthetic posteriors: logits = N(0, sigma) + peak * onehot(planted).  9 (every round so "
                         "far): the aligned path loses ~0.4 per frame after the reference's target boost; 7: ~1.3 (a 1000-frame "
                         "utterance ends below the reference's -1000 sentinel); 3: ~6")
    ap.add_argument("--sigma", type=float, default=1.0, help="noise scale of the synthetic logits")
    ap.add_argument("--tok-div", type=int, default=None, help="--ragged / c5proxy: targets per utterance = frames // this (25; c5proxy: 12)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="N ranks with REAL GPU work on fewer GPUs than ranks (device = LOCAL_RANK %% device_count) over a gloo "
                         "group with the records staged through pinned host memory: correctness of the N > 1 path on one GPU "
                         "(timings are meaningless there)")
    ap.add_argument("--force-group", action="store_true",
                    help="initialise the process group even at world size 1 (exercises the N > 1 code path of the headline mode)")
    args = ap.parse_args()

    # the hardware-queue count is a runtime setting read when HIP initialises (nothing has touched the device yet)
    if args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))

    if args.config == "c2":
        args.batch, args.frames, args.tokens = 256, 600, 20
    if args.config == "c5proxy":   # BASELINE.json configs[4] without the model: segments of <= 30 s, both heads, SIL in the targets
        if (args.tlo, args.thi) == (200, 3000):
            args.tlo, args.thi = 300, 1870
        args.tok_div = args.tok_div or 12
    args.tok_div = args.tok_div or 25
    if args.parity_sample is None:
        args.parity_sample = 512 if args.config == "realtext" else (64 if args.config == "c5proxy" else 256)
    rk = Ranks(args, need_group=(args.config == "c4" or args.force_group))
    if args.ragged:
        ragged_main(args, rk)
    elif args.config == "c4":
        c4_main(args, rk)
    elif args.config == "realtext" and not args.dry_run:
        realtext_main(args, rk)
    elif args.config == "c5proxy" and not args.dry_run:
        c5proxy_main(args, rk)
    elif args.dry_run:
        dry_headline(args, rk)
    else:
        headline_main(args, rk)
    rk.close()  # (not in a `finally`: a rank that failed must not wait in the closing barrier)


if __name__ == "__main__":
    main()
