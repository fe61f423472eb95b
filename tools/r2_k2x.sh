ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/k2x; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in base k2x1 k2x2 k2x4 k2x7; do
  if [ $v = base ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/tools/ubench/dbg/libbfa_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python $ROOT/bench.py --ragged --steps 5 --no-cpu --batch 64 --tlo 2990 --thi 3000 > $OUT/$v.log 2>&1
  echo "== $v"; grep "k_backtrace_wide\|k_dp5" $OUT/$v/*kernel_stats.csv | cut -d, -f1-6 | cut -c1-150
done
