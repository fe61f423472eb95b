#!/bin/bash
# tools/experiments/scripts/r4_lm_sparse.sh -- lane masks taken at every frame (the build) or only where the band moves, for windows of
# >= 1 / >= 2 slots in k_one, k_dp4x and k_mix (variants/libbfa_spall1.so / spall2.so); one box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), (d.get('alignment_only') or {}).get('ms_per_step'))"; }
for lib in spall1 spall2; do BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so timeout 900 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1; done
for rep in 1 2; do for lib in build spall1 spall2; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --ragged --steps 30 2>/dev/null | last | ms "$lib ragged"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 0 2>/dev/null | last | ms "$lib shard4096"
  python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 0 2>/dev/null | last | ms "$lib c4"
  python bench.py --config c2 --steps 50 --warmup 10 --no-cpu --inflight 1 2>/dev/null | last | ms "$lib c2"
  python tools/latency_device.py 2>/dev/null | grep "B=" | sed "s/^/$lib /"
  python tools/latency_mixed.py 2>/dev/null | grep -E "B=(2|16|64):" | sed "s/^/$lib /"
  python tools/experiments/scripts/r4_xwin_lm.py 2>/dev/null | grep "B=" | sed "s/^/$lib /"
done; done
unset BFA_HIP_LIBRARY
