ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for B in 1024 2048 4096; do for mode in normal same normal same; do
  if [ $mode = same ]; then export BFA_BENCH_SAME_INPUT=1; else unset BFA_BENCH_SAME_INPUT; fi
  echo -n "B=$B $mode: "; BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_nos4.so python $ROOT/bench.py --no-cpu --batch $B --steps 200 --warmup 20 2>/dev/null | python $ROOT/tools/ubench/extract.py /dev/stdin
done; done
