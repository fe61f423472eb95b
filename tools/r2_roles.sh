ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in base noprod nocons; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  echo "== $v"
  python bench.py --no-cpu --steps 40 --warmup 5 --inflight 1 | python tools/ubench/extract.py /dev/stdin
  BFA_BENCH_SAME_INPUT=1 python bench.py --no-cpu --steps 40 --warmup 5 --inflight 1 | python tools/ubench/extract.py /dev/stdin
done
