ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for c in 4096 8192 16384 32768; do
  python bench.py --config c4 --steps 4 --chunk $c --parity-sample 32 2>/dev/null > gpurun_out/c4_$c.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/c4_$c.json").read().strip().split("\n")[-1])
print("chunk", $c, "ms", round(d["ms_per_step"],3), "G/s", round(d["value"]/1e9,2), "frac", round(d["roofline"]["frac"],3), "mism", d["parity_sample"]["mismatching_utterances"])
PY
done
