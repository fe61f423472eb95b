ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for n in nostore nos4 nofetch; do for mode in normal same normal same; do
  if [ $mode = same ]; then export BFA_BENCH_SAME_INPUT=1; else unset BFA_BENCH_SAME_INPUT; fi
  echo -n "$n $mode: "; BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$n.so python $ROOT/bench.py --no-cpu --steps 200 --warmup 20 2>/dev/null | python $ROOT/tools/ubench/extract.py /dev/stdin
done; done
