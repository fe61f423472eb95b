# tools/r5_stagger_ab.sh -- the planner's LDS footprint (BFA_PLAN_LDS_FULL = rounds 3-4), its grid (BFA_PLAN_GRID) and staggered
# heads (BFA_STAGGER_HEADS = 0 / 1 / 2) on the real-text step, interleaved on one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "segment or sil or level2 or realtext or golden or pipeline or planner" 2>&1 | tail -2
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
}
for rep in 1 2 3; do
  run "full-lds stagger=0" BFA_PLAN_LDS_FULL=1 BFA_STAGGER_HEADS=0
  run "small-lds stagger=0" BFA_STAGGER_HEADS=0
  run "small-lds stagger=1" BFA_STAGGER_HEADS=1
  run "small-lds stagger=2" BFA_STAGGER_HEADS=2
  run "small-lds stagger=0 grid2048" BFA_STAGGER_HEADS=0 BFA_PLAN_GRID=2048
  run "small-lds stagger=1 grid2048" BFA_STAGGER_HEADS=1 BFA_PLAN_GRID=2048
done
for rep in 1 2; do
  for v in full small; do
    if [ $v = full ]; then export BFA_PLAN_LDS_FULL=1; else unset BFA_PLAN_LDS_FULL; fi
    python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v-lds realtext 3 in flight %.4f ms' % d['ms_per_step'])"
    python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v-lds sil %.4f ms' % d['ms_per_step'])"
  done
done
unset BFA_PLAN_LDS_FULL
for sg in 0 1; do
  export BFA_STAGGER_HEADS=$sg
  echo "== small-lds stagger=$sg"
  bash tools/timeline.sh r5sg$sg 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026"
done > gpurun_out/r5_stagger_timeline2.txt 2>&1
grep "last step\|==" gpurun_out/r5_stagger_timeline2.txt
