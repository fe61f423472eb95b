#!/usr/bin/env python3
"""tools/prof_summary.py <gpurun_out/prof_TAG> -- condense rocprofv3 CSV output (kernel stats + PMC
passes from tools/profile.sh) into a text summary suitable for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    out = []
    st = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
    if st:
        out.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        rows = list(csv.DictReader(open(st[0])))
        for r in rows:
            if "bfa" in r["Name"] or float(r["Percentage"]) > 1.0:
                out.append(f'{r["Name"][:72]:72s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"]) / 1e3:10.1f} '
                           f'min_us={float(r["MinNs"]) / 1e3:9.1f} max_us={float(r["MaxNs"]) / 1e3:9.1f} pct={r["Percentage"]}')
    agg = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "bfa" not in name:
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        out.append("")
        out.append("== PMC counters per dispatch (mean over dispatches) ==")
        for k, cs in agg.items():
            out.append(k[:100])
            for c, v in sorted(cs.items()):
                out.append(f"    {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
