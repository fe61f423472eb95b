#!/bin/bash
# tools/experiments/scripts/r4_soak.sh -- more seeds of the randomized differential soak on the final build (appends to gpurun_out/r4f/soak_more.json)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r4f
for s in ${SOAK_SEEDS:-31 32 33 34 35 36 37 38 39 40 41 42}; do timeout 900 python tests/soak.py 250 $s --record gpurun_out/r4f/soak_more.json 2>&1 | tail -1; done
