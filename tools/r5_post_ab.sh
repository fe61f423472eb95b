# tools/r5_post_ab.sh -- k_postconf: staging windows clipped to the gaps between tuples (build) against the +-8-frame margins of
# rounds 3-4 (variants/libbfa_post_old.so), and its grid (BFA_POST_GRID: a resident k_postconf wave holds ~100 VGPRs), one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "level2 or post or conf or realtext or pipeline or golden" 2>&1 | tail -2
V=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_post_old.so
for rep in 1 2 3; do for lib in old build; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$V; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
done; done
unset BFA_HIP_LIBRARY
for rep in 1 2 3; do for g in 65536 2048 1024; do
  BFA_POST_GRID=$g python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('post_grid=$g headline %.4f ms' % d['ms_per_step'])"
  BFA_POST_GRID=$g python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('post_grid=$g realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  BFA_POST_GRID=$g python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('post_grid=$g realtext inflight1 %.4f ms' % d['ms_per_step'])"
done; done
echo "== build"
bash tools/timeline.sh r5po 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r5_post_timeline.txt
grep "last step\|k_postconf" gpurun_out/r5_post_timeline.txt
for s in 41 42; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
