#!/bin/bash
# tools/experiments/scripts/r4_mix_waves.sh -- k_mix at 6 (the build) / 7 / 5 waves per SIMD after the lane-mask changes; one box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
for rep in 1 2; do for lib in build w7 w5; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --ragged --steps 30 2>/dev/null | last | ms "$lib ragged"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 0 2>/dev/null | last | ms "$lib shard4096"
  python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 0 2>/dev/null | last | ms "$lib c4"
done; done
