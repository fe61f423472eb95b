ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
for v in base r4s; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  echo "== $v"
  python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
  python bench.py --ragged --no-cpu --batch 64 --tlo 1590 --thi 1600 | grep "^{" | cut -c60-150
  python bench.py --ragged --no-cpu --batch 4096 --tlo 1100 --thi 1600 | grep "^{" | cut -c60-150
  python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
  python bench.py --config c4 --steps 4 --parity-sample 128 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
done; done
export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_r4s.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
