cd $GRAFT_REPO_ROOT
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'])"; }
for rep in 1 2; do
for lib in build nopair; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$GRAFT_REPO_ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config realtext --inflight 1 --no-cpu --parity-sample 0 2>/dev/null | j "$lib q4 if1"
  python bench.py --config realtext --inflight 3 --no-cpu --parity-sample 0 2>/dev/null | j "$lib q4 if3"
done; done
