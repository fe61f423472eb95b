#!/bin/bash
# tools/r3_remap.sh -- headline K1 with every XCD on a contiguous range of utterances (the build) against workgroup id = utterance
# (variant noremap), per resident batch, interleaved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { BFA_BENCH_DUMP_K1=1 python bench.py --steps 20 --warmup 5 --no-cpu "$@" 2>&1 | grep "mean per buffer\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('   ms/step %.4f  K1 %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
    else: print('  ', l.strip())"; }
for i in 1 2 3; do
echo "remap"; run
echo "no remap"; BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_noremap.so run
done
