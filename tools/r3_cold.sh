#!/bin/bash
# tools/r3_cold.sh -- headline K1 with the band / window updates laid out as cold branches (variant cold4) against the build, interleaved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'K1 ms', (d.get('roofline') or {}).get('kernel_ms'))"; }
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j "as built"
BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_cold4.so python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j "cold branches"
done
