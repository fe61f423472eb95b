#!/usr/bin/env python3
"""tools/resource_report.py file... -- one line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(stderr of a compile): SGPRs, VGPRs, spills, scratch, occupancy, LDS."""
import re
import subprocess
import sys


def parse(text):
    out = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = blk.split()[0]
        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = name
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).replace("bfa::", "").replace("void ", "")
        dem = re.sub(r"\(.*\)$", "", dem)
        out.append(dict(name=dem, sgpr=g("SGPRs"), vgpr=g("VGPRs"), agpr=g("AGPRs"), sspill=g("SGPRs Spill"),
                        vspill=g("VGPRs Spill"), scratch=g("ScratchSize [bytes/lane]"), occ=g("Occupancy [waves/SIMD]"),
                        lds=g("LDS Size [bytes/block]")))
    return out


if __name__ == "__main__":
    rows = []
    for f in sys.argv[1:]:
        rows += parse(open(f).read())
    print(f"{'kernel':70s} {'sgpr':>5s} {'vgpr':>5s} {'sspill':>6s} {'vspill':>6s} {'scr':>4s} {'occ':>3s} {'lds':>6s}")
    for r in sorted(rows, key=lambda r: r["name"]):
        print(f"{r['name'][:70]:70s} {r['sgpr']:5d} {r['vgpr']:5d} {r['sspill']:6d} {r['vspill']:6d} {r['scratch']:4d} {r['occ']:3d} {r['lds']:6d}")
