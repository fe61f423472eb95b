#!/bin/bash
# tools/r5_rt_half.sh -- timing experiment (WRONG results in the variant): k_dp4_any with half the consumer's arithmetic
# (-DBFA_EXP_HALF_STEP: slot 1 of every lane skipped) against the build -- what could a one-state-per-lane class or two pieces
# per wavefront gain at most?
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for lib in build halfstep; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  echo "== $lib"
  bash tools/timeline.sh rth_$lib 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep "k_dp4_any\|last step\|rror" | cut -c1-120
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>&1 | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'])"
done; done
