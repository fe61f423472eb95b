#!/bin/bash
# tools/r5_uniform_ab.sh -- same box: item records passed through v_readfirstlane before the walks (uniform_item, bfa_types.hpp)
# against the records as loaded: k_mix on the mixed-length workloads (variant nouni), k_backtrace on the headline (nounibt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
NO_C4=${NO_C4:-} bash tools/ab.sh "nouni" 3 2>&1
for rep in 1 2 3; do for lib in build nounibt; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib headline %.4f ms' % d['ms_per_step'], 'walk', d.get('kernels_us'))"
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
done; done
unset BFA_HIP_LIBRARY
for s in 61 62; do timeout 900 python tests/soak.py 100 $s 2>&1 | tail -1; done
