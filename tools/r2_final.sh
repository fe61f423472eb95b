ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for s in 11 12 13 14; do timeout 900 python tests/soak.py 300 $s 2>&1 | tail -1; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; cut -c1-400 gpurun_out/final_bench.json
