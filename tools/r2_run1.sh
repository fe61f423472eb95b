set -x
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r2a/scale_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/scale_tests.log
python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err
python bench.py --inflight 2 --no-cpu > gpurun_out/r2a/bench_inflight2.json 2> gpurun_out/r2a/bench_inflight2.err
python bench.py --inflight 3 --no-cpu > gpurun_out/r2a/bench_inflight3.json 2> gpurun_out/r2a/bench_inflight3.err
python bench.py --config c4 --steps 5 > gpurun_out/r2a/bench_c4.json 2> gpurun_out/r2a/bench_c4.err
python bench.py --config c4 --steps 5 --chunk 8192 > gpurun_out/r2a/bench_c4_8k.json 2> gpurun_out/r2a/bench_c4_8k.err
python bench.py --ragged --no-cpu > gpurun_out/r2a/bench_ragged.json 2>&1
tail -3 gpurun_out/r2a/scale_tests.log
cat gpurun_out/r2a/*.json | cut -c1-1500
