ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ragpmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --ragged --steps 3 --no-cpu"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- $B > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o p -- $B > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/p3 -o p -- $B > $OUT/p3.log 2>&1
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt
grep -A 26 "k_dp5<8\|k_dp4<6, 4, 3" $OUT/summary.txt | head -80
