#!/bin/bash
# tools/r3_place.sh -- does the address of the posterior buffer decide K1's duration?  (per-buffer means of the kernel leg)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { echo "== $*"; BFA_BENCH_DUMP_K1=1 python bench.py --steps 20 --warmup 5 --no-cpu "$@" 2>&1 | grep "posterior buffers\|mean per buffer"; }
run; run
run --place-align 2097152
run --place-align 2097152 --place-offset 64
run --place-align 2097152 --place-offset 256
run --place-align 2097152 --place-offset 4096
run --place-align 2097152 --place-offset 65536
run --place-align 1073741824
run
