set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${1:-y}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL -d $OUT/pmc1 -o p -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_VALU SQ_WAVES_EQ_64 SQ_INSTS_FLAT -d $OUT/pmc2 -o p -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_ATOMIC SQC_DCACHE_INPUT_VALID_READYB SQC_TC_REQ SQC_TC_DATA_WRITE_REQ SQC_TC_STALL -d $OUT/pmc3 -o p -- $BENCH > $OUT/pmc3.log 2>&1
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt
grep -A 30 "k_dp4w" $OUT/summary.txt | head -50
tail -3 $OUT/pmc1.log $OUT/pmc2.log $OUT/pmc3.log
