cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o x -- python -m pytest tests/test_gpu_xwin.py -x -q -m gpu > gpurun_out/r4/xwin_prof.log 2>&1
find /tmp/prof_x -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4/xwin_kernel_stats.csv
cut -d, -f1-4 gpurun_out/r4/xwin_kernel_stats.csv | head -40
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
