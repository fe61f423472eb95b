ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
for v in base nc4; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  echo "== $v"
  python bench.py --ragged --no-cpu --batch 64 --tlo 2990 --thi 3000 | grep "^{" | cut -c60-150
  python bench.py --ragged --no-cpu --batch 2640 --tlo 2400 --thi 3000 | grep "^{" | cut -c60-150
  python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
  python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
  python bench.py --config c4 --steps 4 --parity-sample 128 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
unset BFA_HIP_LIBRARY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
