# tools/r5_pieces_ab.sh -- the pieces' sub-silences in a kernel of their own (k_plan_pieces, one wavefront per piece: build) against
# the previous commit's planner (variants/libbfa_plan_prev.so); phase clocks of the build's planner; one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "segment or sil or level2 or realtext or golden or pipeline or planner" 2>&1 | tail -2
BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_plan_stamps.so python tools/plan_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_plan_stamps_pieces.txt
for rep in 1 2 3; do for lib in plan_prev build; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib sil %.4f ms' % d['ms_per_step'])"
done; done
unset BFA_HIP_LIBRARY
for g in 2048 8192 16384; do
  BFA_PIECES_GRID=$g python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pieces_grid=$g realtext inflight1 %.4f ms' % d['ms_per_step'])"
done
echo "== build"
bash tools/timeline.sh r5pc 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r5_pieces_timeline.txt
grep -v "rocprofv3\|amdgpu.ids" gpurun_out/r5_pieces_timeline.txt
for s in 61 62 63 64; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
