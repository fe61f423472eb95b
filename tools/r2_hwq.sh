ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  echo "== GPU_MAX_HW_QUEUES=$q"
  python bench.py --ragged --no-cpu | grep "^{" | cut -c1-160
  for fl in 1 2; do
    python bench.py --config c4 --global-batch 4096 --steps 12 --inflight $fl --parity-sample 32 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard inflight', $fl, 'ms', round(d['ms_per_step'],3))"
  done
  python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
  python bench.py --no-cpu --steps 20 --warmup 5 --inflight 2 | python tools/ubench/extract.py /dev/stdin
done
