#!/usr/bin/env python3
"""tools/k1_busy.py <kernel_trace.csv> [n_last] -- K1 of the headline under batches in flight: from a rocprofv3 kernel
trace, the last n K1 launches (default 20 = bench.py's reported window): mean launch duration (what --stats averages),
the union of the launch intervals (time during which K1 was running at all) and the busy time per launch that
bench.py's roofline uses."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_dp4w<2, 4, 3, false>" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-n:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
dur = [b - a for a, b in iv]
busy, ca, cb = 0, None, None
for a, b in iv:
    if cb is None or a > cb:
        busy += (cb - ca) if cb is not None else 0
        ca, cb = a, b
    else:
        cb = max(cb, b)
busy += (cb - ca) if cb is not None else 0
alg = 4096 * 1000 * 317
print(f"K1 launches: {len(iv)} (the last {n} of the trace), queues: {sorted(set(r.get('Queue_Id', '?') for r in rows))}")
print(f"mean launch duration     {sum(dur) / len(dur) / 1e3:8.1f} us   (min {min(dur) / 1e3:.1f}, max {max(dur) / 1e3:.1f})")
print(f"first start -> last end  {(iv[-1][1] - iv[0][0]) / 1e3:8.1f} us")
print(f"union of the intervals   {busy / 1e3:8.1f} us   = {busy / len(iv) / 1e3:.1f} us of K1 busy time per launch")
print(f"launches running on average: {sum(dur) / busy:.2f}")
print(f"algorithmic bytes per launch 1.298 GB -> {alg / (busy / len(iv)) :.0f} GB/s of busy time = {alg / (busy / len(iv)) / 8000 * 100:.1f} % of 8 TB/s;"
      f" per launch duration {alg / (sum(dur) / len(dur)):.0f} GB/s = {alg / (sum(dur) / len(dur)) / 8000 * 100:.1f} %")
