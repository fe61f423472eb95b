#!/bin/bash
# tools/r3_remap.sh -- headline K1 with BFA_HINT_UNIFORM_LENGTHS (each XCD on one contiguous eighth of the batch) against
# workgroup id = utterance (--no-uniform-hint), per resident batch, interleaved processes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('   ms/step %.4f  K1 %.4f  per batch %s' % (d['ms_per_step'], r['kernel_ms'], ' '.join('%.4f' % v for v in r['kernel_ms_per_buffer'])))"; }
for i in 1 2 3; do
echo "eighth per XCD"; run
echo "workgroup = utterance"; run --no-uniform-hint
done
