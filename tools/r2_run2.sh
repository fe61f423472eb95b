set -x
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q > gpurun_out/r2b/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/gpu_tests.log
tail -40 gpurun_out/r2b/gpu_tests.log
timeout 600 python tests/soak.py 300 778 > gpurun_out/r2b/soak.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/soak.log
tail -5 gpurun_out/r2b/soak.log
