#!/bin/bash
# tools/r5_uniform_ab.sh -- same box: k_mix's walk on an item record loaded again behind the fence and passed through
# v_readfirstlane (uniform_item, bfa_types.hpp; BFA_MIX_UNIFORM_WALK=1, the build) against the record as loaded at the top
# (variant nouniw) on the mixed-length workloads; first the tests that cover the kernels touched
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_gpu_mix.py tests/test_gpu_xwin.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab.sh "nouniw" 3 2>&1
for s in 71 72 73; do timeout 900 python tests/soak.py 100 $s 2>&1 | tail -1; done
