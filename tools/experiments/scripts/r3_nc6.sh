ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_nc6_3.so
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_consumer or mixed_length or soak_slice" 2>&1 | tail -2
unset BFA_HIP_LIBRARY
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 %.4f' % d['ms_per_step'], d.get('status_ok', (d.get('parity_sample') or {}).get('mismatching_utterances')), end='  ')"; }
for i in 1 2; do for v in build nc6_3; do
  if [ $v = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so; fi
  echo -n "$v: "
  python bench.py --ragged --steps 30 2>/dev/null | j ragged
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 2>/dev/null | j "c4 shard"
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 64 2>/dev/null | j "c4"
  echo
done; done
