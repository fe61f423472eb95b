set -x
mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -q -x > gpurun_out/r2c/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/gpu_tests.log
tail -5 gpurun_out/r2c/gpu_tests.log
python bench.py > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err
python bench.py --inflight 2 --no-cpu > gpurun_out/r2c/bench_inflight2.json 2>/dev/null
python bench.py --classes 17 --no-cpu > gpurun_out/r2c/bench_c17.json 2>/dev/null
python bench.py --ragged --no-cpu > gpurun_out/r2c/bench_ragged.json 2>/dev/null
python bench.py --config c4 --steps 5 --chunk 8192 > gpurun_out/r2c/bench_c4.json 2>/dev/null
python tools/ubench/extract.py gpurun_out/r2c/bench_default.json gpurun_out/r2c/bench_inflight2.json gpurun_out/r2c/bench_c17.json
cat gpurun_out/r2c/bench_ragged.json; cut -c1-400 gpurun_out/r2c/bench_c4.json
timeout 900 python tests/soak.py 200 901 > gpurun_out/r2c/soak.log 2>&1; tail -2 gpurun_out/r2c/soak.log
