ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for fl in 1 2 3; do
  python bench.py --config c4 --global-batch 4096 --steps 12 --inflight $fl --parity-sample 64 2>/dev/null | grep "^{" > gpurun_out/c4fl_$fl.json
  python bench.py --config c4 --steps 6 --inflight $fl --parity-sample 64 2>/dev/null | grep "^{" > gpurun_out/c4flbig_$fl.json
done
