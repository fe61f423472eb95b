# tools/r5_slices_ab.sh -- bfa_align_heads in ranges of utterances (BFA_HEAD_SLICES = 1 / 2 / 3), one box, interleaved
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
for rep in 1 2 3; do for n in 1 2 3; do
  BFA_HEAD_SLICES=$n python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 256 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slices=$n realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'], d['parity'].get('confidence_beyond_1e-4'))"
done
BFA_HEAD_SLICES=2 python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --hw-queues 8 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slices=2 8 queues realtext inflight1 %.4f ms' % d['ms_per_step'])"
BFA_HEAD_SLICES=2 BFA_SLICE_4STREAMS=1 python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --hw-queues 8 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slices=2 4streams 8 queues realtext inflight1 %.4f ms' % d['ms_per_step'])"
done
export BFA_HEAD_SLICES=2
echo "== slices=2"
bash tools/timeline.sh r5sl2 4 python $PWD/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026" > gpurun_out/r5_slices_timeline.txt
grep -v "rocprofv3\|amdgpu.ids" gpurun_out/r5_slices_timeline.txt
