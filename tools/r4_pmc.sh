#!/bin/bash
# tools/r4_pmc.sh -- VALU issue cost microbenchmark + PMC passes of one unsorted mixed-length call and of one rank's C4 shard
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
mkdir -p gpurun_out/r4
./tools/ubench/bin/valu_rates > gpurun_out/r4/valu_rates.txt 2>&1; cat gpurun_out/r4/valu_rates.txt
bash tools/r3_pmc.sh ragged python $ROOT/bench.py --ragged --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
grep -A18 "^k_dp4x<4\|^k_dp4x<3\|^k_dp4w_any\|^k_dp4x_tail<4\|^k_backtrace(" gpurun_out/pmc_ragged/summary.txt
