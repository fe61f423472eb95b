ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
for v in base ahead2 ahead2w8; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  echo "== $v"
  python bench.py --no-cpu --steps 40 --warmup 5 --inflight 1 | python tools/ubench/extract.py /dev/stdin
  python bench.py --no-cpu --steps 40 --warmup 5 | python tools/ubench/extract.py /dev/stdin
done; done
export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_ahead2.so
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "c3 or c2" 2>&1 | tail -2
