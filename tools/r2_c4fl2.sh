ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for q in 4 8; do for f in 1 2 3; do for ch in 16384 8192; do
  echo "queues $q inflight $f chunk $ch"
  GPU_MAX_HW_QUEUES=$q python bench.py --config c4 --steps 6 --inflight $f --chunk $ch --parity-sample 64 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],3), 'G frames/s', round(d['value']/1e9,2), d['parity_sample']['mismatching_utterances'])"
done; done; done
