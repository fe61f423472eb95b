ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do for n in 0 1; do
  echo "== BFA_NARROW_LANES=$n"
  BFA_NARROW_LANES=$n python bench.py --ragged --no-cpu | grep "^{" | cut -c60-150
  BFA_NARROW_LANES=$n python bench.py --config c4 --global-batch 4096 --steps 12 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 shard ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
  BFA_NARROW_LANES=$n python bench.py --config c4 --steps 6 --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   c4 N=1 ms', round(d['ms_per_step'],3), d['parity_sample']['mismatching_utterances'])"
  BFA_NARROW_LANES=$n python tests/sil_time.py | tail -1
done; done

