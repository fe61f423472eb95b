# tools/r5_postw_ab.sh -- k_postconf at 6 / 8 waves per SIMD (80 / 64 VGPRs, with spills) and with fewer workgroups, on the headline
# step (four batches in flight: the confidence pass of one batch runs beside K1 of the next) and on the real-text step; one box
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "level2 or post or conf" 2>&1 | tail -1
for rep in 1 2 3; do
  for lib in build post_w6 post_w8; do
    if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib headline %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'] if 'parity' in d else '')"
    python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 128 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext inflight1 %.4f ms' % d['ms_per_step'], d['parity']['mismatching_utterances'])"
    python bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib realtext 3 in flight %.4f ms' % d['ms_per_step'])"
  done
  unset BFA_HIP_LIBRARY
  for g in 1024 512 256; do
    BFA_POST_GRID=$g python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('post_grid=$g headline %.4f ms' % d['ms_per_step'])"
  done
done
