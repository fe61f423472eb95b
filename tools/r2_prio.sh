ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
echo "== solo classes"
python bench.py --ragged --no-cpu --batch 880 --tlo 2400 --thi 3000 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 1200 --tlo 1600 --thi 2400 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 2000 --tlo 200 --thi 1600 | grep "^{" | cut -c1-140
python bench.py --ragged --no-cpu --batch 64 --tlo 2990 --thi 3000 | grep "^{" | cut -c1-140
for t in 0 1 2 5 6 8 9 10 13 14 0; do
  export BFA_TUNE=$t
  echo "== BFA_TUNE=$t"
  python bench.py --ragged --no-cpu | grep "^{" | cut -c1-140
  python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 32 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d.get('parity_sample'))"
done
for t in 0 1 2; do
  export BFA_TUNE=$t
  echo "== headline BFA_TUNE=$t"
  python bench.py --no-cpu --steps 20 --warmup 5 | python tools/ubench/extract.py /dev/stdin
done
