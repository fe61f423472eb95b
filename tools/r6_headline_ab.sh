cd $GRAFT_REPO_ROOT
ms() { python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'noconf %.4f' % d['alignment_only']['ms_per_step'], 'K1 %.4f' % d['roofline']['kernel_ms'])"; }
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "lazy streams "
  BFA_BENCH_PRECREATE=1 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "aux only     "
  BFA_BENCH_PRECREATE=2 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "head streams "
  BFA_BENCH_PRECREATE=3 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "both         "
  (cd tools/ubench/bin/r05tree && python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | ms "round-5 tree ")
done
