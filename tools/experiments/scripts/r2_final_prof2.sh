ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r02
bash $ROOT/tools/profile.sh r02 > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt
head -14 $OUT/summary.txt
python $ROOT/tools/k1_busy.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 20 > $OUT/k1_busy.txt; cat $OUT/k1_busy.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --settle-ms 60 --inflight 1 > $OUT/trace1.log 2>&1
grep "k_dp4w<2, 4, 3, false>\|k_backtrace\|k_plan\|k_dp4_redo" $OUT/trace1/*kernel_stats.csv | cut -d, -f1-8
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu > $OUT/bench_inflight1.json 2>/dev/null
python tools/ubench/extract.py $OUT/bench.json $OUT/bench_inflight1.json
bash tools/r2_silprof.sh > $OUT/sil.txt 2>&1; cat $OUT/sil.txt
bash tools/r2_ragprof.sh > $OUT/rag.txt 2>&1; tail -20 $OUT/rag.txt
python bench.py --ragged --no-cpu | grep "^{" | cut -c1-200
python bench.py --config c4 --steps 6 --parity-sample 256 2>/dev/null | grep "^{" > $OUT/c4.json; python -c "
import json; d=json.loads(open('$OUT/c4.json').read()); print('c4 N=1', d['ms_per_step'], d['value'], d.get('parity_sample'))"
python tools/pipeline_time.py 4096 1 2>&1 | tail -2
