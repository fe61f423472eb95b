#!/bin/bash
# tools/r5_lpt_ab.sh -- same box: k_dp4_any taking the DP pieces longest first (AlignArgs::piece_list, the build) against item
# order (variant nolpt) on the real-text step and the silence-anchored headline shape; first the tests of the mode, last the
# per-item clocks of the new order (tools/any_stamps.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for lib in build nolpt; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib sil %.4f ms' % d['ms_per_step'])"
done; done
unset BFA_HIP_LIBRARY
bash tools/timeline.sh r5lpt 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^{\|^W2" | cut -c1-120
BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_any_stamps.so timeout 300 python tools/any_stamps.py 2>&1 | grep -v amdgpu.ids
for s in 81 82 83; do timeout 900 python tests/soak.py 100 $s 2>&1 | tail -1; done
