# tools/r5_plan_ab.sh -- the planner of the silence-anchored mode, round 5: P(SIL) staged in LDS, list appends batched, LDS arrays
# sized by the batch, eight planners per CU (the build) against the planner of rounds 3-4 (variants/libbfa_plan_old.so: the
# previous commit's bfa_segment.hip), interleaved on one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
V=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_plan_old.so
for rep in 1 2 3; do for lib in old build; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$V; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib sil %.4f ms' % d['ms_per_step'])"
done; done
for lib in old build; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$V; fi
  echo "== $lib"
  bash tools/timeline.sh r5pl$lib 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026"
done > gpurun_out/r5_plan_timeline.txt 2>&1
unset BFA_HIP_LIBRARY
grep "last step\|==\|k_plan_seg" gpurun_out/r5_plan_timeline.txt
bash tools/pmc.sh r5pl_rt python $PWD/bench.py --config realtext --steps 3 --warmup 1 --settle-ms 0 --min-timed-steps 3 --parity-sample 0 --inflight 1 > /dev/null 2>&1
cp gpurun_out/pmc_r5pl_rt/summary.txt gpurun_out/r5_plan_pmc.txt; grep -A22 "^k_plan_seg" gpurun_out/r5_plan_pmc.txt | head -24
for s in 31 32 33; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
