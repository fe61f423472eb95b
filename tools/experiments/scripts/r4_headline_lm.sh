#!/bin/bash
# tools/experiments/scripts/r4_headline_lm.sh -- the headline's fast-window kernel (k_dp4w<2>) with the scalar band masks (the build), lane
# masks at every frame (variants/libbfa_winlm1.so) or where the band moves (winlm2.so); one box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'K1', round(d['roofline']['kernel_ms'],4), 'align-only', round((d.get('alignment_only') or {}).get('ms_per_step') or 0,4))"; }
BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_winlm2.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do for lib in build winlm1 winlm2; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | ms "$lib headline"
  python bench.py --steps 20 --warmup 5 --no-cpu --inflight 1 --no-confidences 2>/dev/null | last | ms "$lib inflight1-align-only"
done; done
