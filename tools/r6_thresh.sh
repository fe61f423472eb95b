#!/bin/bash
# tools/r6_thresh.sh -- BFA_OPT_WIDE_ANY_MAX_BATCH: from how many utterances on does the full-batch layout of a silence-anchored call
# (k_mix + wide exact windows + k_dp5_any by class set) beat the small-call layout (everything wide in one k_dp5_any launch)?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os, sys, subprocess
for wam in [int(v) for v in os.environ.get("WAMS", "256,8").split(",")]:
    for peak in (9, 5):
        env = dict(os.environ, BFA_BS=os.environ.get("BSS", "16,32,64,128,256"), BFA_PEAK=str(peak), BFA_WIDE_ANY_MAX=str(wam))
        out = subprocess.run([sys.executable, "tools/latency_realtext.py"], env=env, capture_output=True, text=True).stdout
        for l in out.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                print("wide_any_max=%d peak %d B %d device ms %.3f" % (wam, peak, d["B"], d["device_ms_back_to_back"]), flush=True)
PY
