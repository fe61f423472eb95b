# tools/r5_k1prio_ab.sh -- different wave priorities for the DP consumers of the two heads' k_dp4_any (grp_first: group head 3, phoneme
# head 1; ph_first: the reverse; build: both 3): one head's K1 finishes early and its post-DP gathers run beside the other head's K1; one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
for rep in 1 2 3; do for lib in build grp_first ph_first; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
done; done
for lib in grp_first ph_first; do
  export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so
  echo "== $lib"
  bash tools/timeline.sh r5k1$lib 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026\|rocprofv3\|amdgpu.ids"
done > gpurun_out/r5_k1prio_timeline.txt 2>&1
cat gpurun_out/r5_k1prio_timeline.txt
