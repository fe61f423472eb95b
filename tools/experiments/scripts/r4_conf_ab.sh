#!/bin/bash
# tools/experiments/scripts/r4_conf_ab.sh -- headline step with the confidence pass through k_postconf (the build) or through the
# tuple-per-lane k_conf (libbfa_conftuple.so: bfa_capi.cpp with -DBFA_CONF_TUPLE_PER_LANE), same box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], 'alignment only %.4f' % d['alignment_only']['ms_per_step'], 'conf call %.4f' % d['confidence_pass_ms'])"; }
for rep in 1 2 3; do for lib in build conftuple; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j $lib
done; done
unset BFA_HIP_LIBRARY
python __graft_entry__.py smoke 2>&1 | tail -1
