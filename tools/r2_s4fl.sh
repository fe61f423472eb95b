ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2 3; do for v in base nos4; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  echo "== $v"; python bench.py --no-cpu --steps 40 --warmup 5 | python tools/ubench/extract.py /dev/stdin
done; done
