#!/usr/bin/env python3
"""tools/time_reference.py -- CPU-baseline provenance: time the REAL reference
(/root/reference/bournemouth_aligner/forced_alignment.py:856-910, AlignmentUtils.decode_alignments, imported by file
path through tests/refload.py) in the BUILD container on utterances of the BASELINE.json shapes, and write
profiles/reference_cpu_baseline.json.  The reference never ships to the GPU node, so this record is what bench.py
echoes as `reference_cpu_baseline` ("build container, not this node") beside the same-node C port (`cpu_baseline`).

  python tools/time_reference.py [--utts 32] [--procs 8]

Per shape: (a) one process, torch.set_num_threads(1), `--utts` utterances through one decode_alignments call;
(b) a pool of `--procs` processes, each one thread, the same utterances split between them.  The oracle (C port) is
timed on the same utterances in the same process for the port/reference ratio, and its outputs are compared with the
reference's (this doubles as a full-shape parity check of the oracle).
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = {"C2": (600, 20, 1002), "C3": (1000, 40, 1003)}  # T, S, seed


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def _make(T, S, seed, n):
    from tools.synth import synth_batch
    lp, tk = synth_batch(n, T, S, 67, seed, "cpu")
    return lp, tk.to(torch.int64)


def _run_ref(lp, tk, T, S):
    import refload
    fa = refload.forced_alignment()
    au = fa.AlignmentUtils(66, 0)  # reference defaults: anchors 10, ignore_noise, truly_forced
    n = lp.shape[0]
    t0 = time.perf_counter()
    out = au.decode_alignments(lp, tk, torch.full((n,), T), torch.full((n,), S))
    return time.perf_counter() - t0, out


def _worker(args):
    T, S, seed, n, lo, hi = args
    torch.set_num_threads(1)
    lp, tk = _make(T, S, seed, n)
    dt, _ = _run_ref(lp[lo:hi], tk[lo:hi], T, S)
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=32)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu_baseline.json"))
    args = ap.parse_args()
    import refload
    assert refload.available(), "/root/reference is not present: this script runs in the build container only"
    from oracle import oracle as ora
    torch.set_num_threads(1)
    rec = {"what": "AlignmentUtils.decode_alignments of the reference (forced_alignment.py:856-910), CPU, float32",
           "cpu_model": _cpu_model(), "logical_cores": os.cpu_count(), "affinity_cores": len(os.sched_getaffinity(0)),
           "torch": torch.__version__, "python": platform.python_version(), "unit": "aligned frames/s", "shapes": {}}
    for name, (T, S, seed) in SHAPES.items():
        n = args.utts
        lp, tk = _make(T, S, seed, n)
        _run_ref(lp[:1], tk[:1], T, S)  # warm (imports, first-call allocations)
        dt1, ref = _run_ref(lp, tk, T, S)
        o0 = time.perf_counter()
        exp = ora.decode_alignments(lp.numpy(), tk.numpy(), [T] * n, [S] * n, ora.make_params(66, 0), seg_cap=S + 2)
        dto = time.perf_counter() - o0
        mism = sum(1 for b in range(n) if ora.segments_as_lists(exp)[b] != [tuple(int(v) for v in x) for x in ref[b]])
        per = (n + args.procs - 1) // args.procs
        jobs = [(T, S, seed, n, k * per, min(n, (k + 1) * per)) for k in range(args.procs) if k * per < n]
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            pool.map(_worker, [(T, S, seed, 1, 0, 1)] * len(jobs))  # warm the workers
            p0 = time.perf_counter()
            pool.map(_worker, jobs)
            dtp = time.perf_counter() - p0
        # (the pool figure includes each worker re-synthesising the inputs; that is milliseconds against seconds of DP)
        rec["shapes"][name] = {
            "T": T, "S": S, "C": 67, "utterances": n,
            "one_process_one_thread": {"seconds": dt1, "frames_per_s": n * T / dt1},
            "process_pool": {"processes": len(jobs), "seconds": dtp, "frames_per_s": n * T / dtp},
            "c_port_one_thread": {"seconds": dto, "frames_per_s": n * T / dto, "port_over_reference": dt1 / dto},
            "oracle_vs_reference_mismatching_utterances": mism,
        }
        print(name, json.dumps(rec["shapes"][name]), flush=True)
    json.dump(rec, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
