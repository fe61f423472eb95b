#!/bin/bash
# tools/r4_headline_ab.sh -- same box, interleaved: the headline with the stride-4 step from frame 0 (the build) and with the
# general step until the window has left state 0 (libbfa_s4off.so = -DBFA_S4_BASE0=0, the round-3 behaviour); PMC of the build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 ms/step %.4f' % d['ms_per_step'], 'alignment only %.4f' % ((d.get('alignment_only') or {}).get('ms_per_step') or 0), 'K1 %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'per buffer', ' '.join('%.4f' % v for v in r['kernel_ms_per_buffer']))"; }
for rep in 1 2 3; do
  for lib in build s4off; do
    if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j $lib
  done
done
unset BFA_HIP_LIBRARY
bash tools/pmc.sh headline python $ROOT/bench.py --inflight 1 --steps 20 --warmup 5 --no-cpu --no-confidences > /dev/null 2>&1
grep -A28 "^k_dp4w<2" gpurun_out/pmc_headline/summary.txt
