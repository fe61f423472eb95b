#!/bin/bash
# tools/r3_pitch.sh -- A/B of the posterior row pitch: dense 67-float rows (268 B) against rows padded
# to 68 / 72 / 80 floats, interleaved, kernel leg (one batch in flight) and whole step of bench.py.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT; cd $ROOT
for rep in 1 2 3; do
  for p in 0 68 72 80; do
    python bench.py --steps 20 --warmup 5 --no-cpu --row-pitch $p --kernel-leg-steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('pitch', d['config']['row_pitch_floats'], 'step_ms %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'median %.4f' % r['kernel_ms_stats']['median'], 'leg_step_ms %.4f' % r['kernel_leg_ms_per_step'])"
  done
done | tee $OUT/pitch_ab.txt
