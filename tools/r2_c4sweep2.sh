ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for c in 512 1024 2048 4096; do
  python bench.py --config c4 --global-batch 4096 --steps 10 --chunk $c --parity-sample 32 2>/dev/null | grep "^{" > gpurun_out/c4s_$c.json
done
