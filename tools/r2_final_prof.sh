ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/profile.sh r02 > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $ROOT/gpurun_out/prof_r02 > $ROOT/gpurun_out/prof_r02/summary.txt
head -12 $ROOT/gpurun_out/prof_r02/summary.txt
grep -A 30 "k_dp4w<2, 4, 3, false>" $ROOT/gpurun_out/prof_r02/summary.txt | grep -E "FETCH|WRITE|GRBM|INSTS_VALU|LDS_BANK|LDS_IDX" 
cd $ROOT; BFA_BENCH_DUMP_K1=1 python bench.py --no-cpu --steps 60 --warmup 1 --settle-ms 0 2> gpurun_out/prof_r02/k1_series.txt > /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/prof_r02/bench.json 2>/dev/null
cut -c1-600 gpurun_out/prof_r02/bench.json
