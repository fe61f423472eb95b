# tools/r5_pair_ab.sh -- K0 of the 67-column head with two lanes per row (k_silprob_pair: build) against the four-lanes-per-row block
# (variants/libbfa_pair0.so: -DBFA_SILPROB_PAIR=0); parity on every silence-anchored / two-head test first; one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "segment or sil or level2 or realtext or golden or pipeline or planner or heads or group or narrow or minprob" 2>&1 | tail -2
for rep in 1 2 3; do for lib in pair0 build; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 256 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'], d['parity']['confidence_beyond_1e-4'])"
  python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  python tests/sil_time.py 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib sil %.4f ms' % d['ms_per_step'])"
done; done
unset BFA_HIP_LIBRARY
echo "== build"
bash tools/timeline.sh r5pair 2 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026\|rocprofv3\|amdgpu.ids" > gpurun_out/r5_pair_timeline.txt
cat gpurun_out/r5_pair_timeline.txt
for s in 91 92 93 94; do timeout 600 python tests/soak.py 100 $s 2>&1 | tail -1; done
