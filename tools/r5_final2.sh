#!/bin/bash
# tools/r5_final2.sh -- after the last kernel change of the round (K0 of the 17-column head, one lane per row): GPU tests + smoke, the
# headline line once more, and everything the real-text rows of README / DESIGN quote, into gpurun_out/r5g/ (-> profiles/r05_realtext*)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5g; mkdir -p $OUT; cd $ROOT
last() { grep "^{" | tail -1; }
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py --steps 20 --warmup 5 2>/dev/null | last > $OUT/bench.json
python bench.py --config realtext --steps 20 --warmup 5 2>/dev/null | last > $OUT/realtext.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1.json
python bench.py --config realtext --steps 20 --warmup 5 --inflight 1 --hw-queues 8 --parity-sample 0 2>/dev/null | last > $OUT/realtext_inflight1_q8.json
python tests/sil_time.py 2>/dev/null | last > $OUT/sil.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_realtext -o t -- python $ROOT/bench.py --config realtext --steps 20 --warmup 5 --parity-sample 0 > $OUT/st_realtext.log 2>&1
cd $ROOT
cp $(find $OUT/st_realtext -name "*kernel_stats.csv" | head -1) $OUT/st_realtext_kernel_stats.csv 2>/dev/null; rm -rf $OUT/st_realtext
bash tools/timeline.sh r5g_realtext 2 python $ROOT/bench.py --config realtext --steps 5 --warmup 2 --settle-ms 0 --min-timed-steps 5 --parity-sample 0 --inflight 1 2>&1 | grep -v "^W2026\|rocprofv3\|amdgpu.ids" > $OUT/realtext_timeline.txt
bash tools/pmc.sh r5g_rt python $ROOT/bench.py --config realtext --steps 3 --warmup 1 --settle-ms 0 --min-timed-steps 3 --parity-sample 0 --inflight 1 > /dev/null 2>&1
cp gpurun_out/pmc_r5g_rt/summary.txt $OUT/realtext_pmc.txt
for s in 81 82 83 84; do timeout 600 python tests/soak.py 150 $s --record $OUT/soak.json 2>&1 | tail -1; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d.get("ms_per_step",0),4), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel_ms"), d.get("total_utterances"), d.get("total_mismatches"))
    except Exception as e: print(f, e)
PY
grep "k_silprob\|SQ_INSTS_VALU" $OUT/realtext_pmc.txt | grep -A1 "k_silprob" | head -8
