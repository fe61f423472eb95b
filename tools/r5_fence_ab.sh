#!/bin/bash
# tools/r5_fence_ab.sh -- same box: (1) workgroup- against agent-scope fence in front of k_mix's walk on the mixed-length
# workloads; (2) the piece walks fused into k_dp4_any against the separate k_backtrace launch on the silence-anchored path
cd ${GRAFT_REPO_ROOT:-/root/repo}
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab.sh "fence_agent" 2 2>&1
for rep in 1 2; do for lib in build any_old; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib sil %.4f ms' % d['ms_per_step'])"
done; done
unset BFA_HIP_LIBRARY
bash tools/timeline.sh r5rt3 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 > gpurun_out/r5_realtext_timeline2.txt 2>&1; head -17 gpurun_out/r5_realtext_timeline2.txt
for s in 21 22 51 52; do timeout 900 python tests/soak.py 100 $s 2>&1 | tail -1; done
