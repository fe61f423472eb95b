#!/bin/bash
# round 6: the bench lines (headline, c5proxy at three sharpnesses, realtext), rocprofv3 stats of the c5proxy step, the new bench test
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
TAG=${1:-r6k}
mkdir -p gpurun_out/$TAG
for pk in 9 6 3; do python bench.py --config c5proxy --peak $pk --steps 10 --warmup 3 --min-timed-steps 30 > gpurun_out/$TAG/c5proxy_p$pk.json 2>> gpurun_out/$TAG/err.log; done
python bench.py --config c5proxy --inflight 3 --steps 10 --warmup 3 --min-timed-steps 30 > gpurun_out/$TAG/c5proxy_p9_inflight3.json 2>> gpurun_out/$TAG/err.log
for pk in 7 5; do python bench.py --peak $pk --steps 20 --warmup 5 --no-cpu > gpurun_out/$TAG/headline_p$pk.json 2>> gpurun_out/$TAG/err.log; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$TAG/*.json")):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'], 'value %.3g' % d['value'], 'frac %.3f' % d['roofline']['frac'], d.get('parity'), (d.get('softness') or {}).get('last_call'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG/prof_c5 -o t -- python $ROOT/bench.py --config c5proxy --steps 10 --warmup 3 --min-timed-steps 20 --parity-sample 0 > $ROOT/gpurun_out/$TAG/prof_c5.log 2>&1
cd $ROOT; cp $(find gpurun_out/$TAG/prof_c5 -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/c5proxy_kernel_stats.csv; head -14 gpurun_out/$TAG/c5proxy_kernel_stats.csv | cut -c1-150
python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "c5proxy" 2>&1 | tail -3
