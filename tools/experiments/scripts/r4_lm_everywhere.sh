#!/bin/bash
# tools/experiments/scripts/r4_lm_everywhere.sh -- lane-computed band masks beyond k_one: k_mix (the build), the fast-window class
# kernels (variants/libbfa_winlm.so: the headline's k_dp4w), the exact-window class kernels (libbfa_xwinlm.so: Rw 6 / 8, lone
# utterances); one box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('kernel_ms'))"; }
timeout 900 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
  python bench.py --ragged --steps 30 2>/dev/null | last | ms ragged
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 0 2>/dev/null | last | ms shard4096
  python bench.py --config c4 --steps 8 --warmup 2 2>/dev/null | last | ms c4
  for lib in build winlm; do
    if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | ms "headline $lib"
    python bench.py --steps 20 --warmup 5 --no-cpu --inflight 1 --no-confidences 2>/dev/null | last | ms "headline inflight1 align-only $lib"
  done
  for lib in build xwinlm; do
    if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
    python tools/one_long.py 2>/dev/null | tail -3 | sed "s/^/$lib /"
  done
  unset BFA_HIP_LIBRARY
done
