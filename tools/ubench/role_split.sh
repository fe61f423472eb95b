#!/bin/bash
# Role-time experiment for K1 (producer = softmax wave, consumer = DP wave): builds two extra copies of the
# library with one role stubbed out (results are wrong, only the kernel time is of interest).
# Run here (cross-compile), then time on the GPU box with tools/ubench/role_time.py.
set -eu
cd "$(dirname "$0")/../../bournemouth-forced-aligner_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I. -I../../include"
OUT=../../tools/ubench/dbg
mkdir -p $OUT
for v in NOPRODUCE NOCONSUME; do
  /opt/rocm/bin/hipcc $FLAGS -DBFA_DBG_$v -c bfa_dp_nk5.hip -o $OUT/nk5_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libbfa_$v.so build/bfa_kernels.hip.o build/bfa_dp_nk2.hip.o $OUT/nk5_$v.o \
      build/bfa_dp_nk8.hip.o build/bfa_backtrace.hip.o build/bfa_segment.hip.o build/bfa_post.hip.o build/bfa_capi.cpp.o
done
