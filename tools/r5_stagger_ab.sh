# tools/r5_stagger_ab.sh -- staggered heads (BFA_STAGGER_HEADS = 0 / 1) x planner wave priority (0: build, 3: variant), one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
V=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_plan_prio3.so
run() { name=$1; shift
  env "$@" python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
}
for rep in 1 2 3; do
  run "stagger=0 prio0" BFA_STAGGER_HEADS=0
  run "stagger=1 prio0" BFA_STAGGER_HEADS=1
  run "stagger=0 prio3" BFA_STAGGER_HEADS=0 BFA_HIP_LIBRARY=$V
  run "stagger=1 prio3" BFA_STAGGER_HEADS=1 BFA_HIP_LIBRARY=$V
  run "stagger=2 prio3" BFA_STAGGER_HEADS=2 BFA_HIP_LIBRARY=$V
done
export BFA_STAGGER_HEADS=1 BFA_HIP_LIBRARY=$V
echo "== stagger=1 prio3"
bash tools/timeline.sh r5sg1p3 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r5_stagger_timeline4.txt 2>&1
grep -v "rocprofv3\|amdgpu.ids" gpurun_out/r5_stagger_timeline4.txt
