#!/bin/bash
# tools/r4_pmc_uniform.sh -- instruction counts per frame of ONE class: uniform batches (the per-class kernels run alone)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for shape in "1024 2600 104" "1024 1800 72" "1024 1000 40"; do
  set -- $shape
  bash tools/r3_pmc.sh u$2 python $ROOT/bench.py --batch $1 --frames $2 --tokens $3 --steps 2 --warmup 1 --no-cpu --min-timed-steps 2 --kernel-leg-steps 2 > /dev/null 2>&1
  echo "== B=$1 T=$2 S=$3"
  grep -A18 "^k_dp4\|^k_backtrace(" gpurun_out/pmc_u$2/summary.txt | grep "^k_\|SQ_INSTS_VALU\|SQ_INSTS_SALU\|SQ_WAVES \|GRBM_GUI"
done
