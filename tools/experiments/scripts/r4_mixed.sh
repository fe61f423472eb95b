#!/bin/bash
# tools/r4_mixed.sh -- the mixed-length workloads (one rank's C4 shard, C4 at N = 1, one unsorted call) and the headline on
# the current build, with the kernel statistics of the C4 shard
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
last() { grep "^{" | tail -1; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'parity', (d.get('parity_sample') or {}).get('mismatching_utterances', d.get('parity_mismatching_utterances')), 'frac', (d.get('roofline') or {}).get('frac'))"; }
python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 2>/dev/null | last > $OUT/c4_shard4096$1.json; show shard < $OUT/c4_shard4096$1.json
python bench.py --ragged --steps 30 2>/dev/null | last > $OUT/ragged$1.json; show ragged < $OUT/ragged$1.json
python bench.py --config c4 --steps 10 --warmup 2 2>/dev/null | last > $OUT/c4$1.json; show c4 < $OUT/c4$1.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last > $OUT/bench$1.json; show headline < $OUT/bench$1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_shard -o t -- python $ROOT/bench.py --config c4 --global-batch 4096 --steps 10 --warmup 2 --parity-sample 0 > $OUT/st_shard.log 2>&1
cp $(find $OUT/st_shard -name "*kernel_stats.csv" | head -1) $OUT/shard_kernel_stats$1.csv; rm -rf $OUT/st_shard
cut -d, -f1-4 $OUT/shard_kernel_stats$1.csv | head -24
