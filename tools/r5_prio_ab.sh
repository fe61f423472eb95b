# tools/r5_prio_ab.sh -- wave priorities: the planner (p = 0 build, 2, 3) against the DP consumers of k_dp4_any (c = 3 build, 2, 1), one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
for rep in 1 2 3; do for lib in build p0c2 p3c2 p3c1 p2c1; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
done; done
for lib in build p3c1; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  echo "== $lib"
  bash tools/timeline.sh r5pr$lib 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026\|rocprofv3\|amdgpu.ids"
done > gpurun_out/r5_prio_timeline.txt 2>&1
grep "==\|k_plan_seg\|k_dp4_any\|last step" gpurun_out/r5_prio_timeline.txt
