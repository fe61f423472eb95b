#!/bin/bash
# Role-time experiment for K1 (producer = softmax wave, consumer = DP wave): builds two extra copies of the
# library with one role stubbed out (results are wrong, only the kernel time is of interest).
# Run here (cross-compile), then time on the GPU box with tools/ubench/role_time.py.
set -eu
cd "$(dirname "$0")/../../bournemouth-forced-aligner_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I. -I../../include"
OUT=../../tools/ubench/dbg
mkdir -p $OUT
NK=${1:-nk5}   # nk5: the 67-class head, nk2: the 17-class head
for v in NOPRODUCE NOCONSUME; do
  /opt/rocm/bin/hipcc $FLAGS -DBFA_DBG_$v -c bfa_dp_$NK.hip -o $OUT/${NK}_$v.o &
done
wait
for v in NOPRODUCE NOCONSUME; do
  OBJS=""
  for f in bfa_kernels.hip bfa_dp_nk5_p3.hip bfa_dp_nk2_p3.hip bfa_dp_nk5_p2.hip bfa_dp_nk2_p2.hip bfa_dp_nk5_p4.hip bfa_dp_nk2_p4.hip bfa_dp_nk5_p5.hip bfa_dp_nk2_p5.hip bfa_dp_nk2.hip bfa_dp_nk5.hip bfa_dp_nk8.hip bfa_backtrace.hip bfa_segment.hip bfa_post.hip bfa_stitch.hip bfa_capi.cpp; do
    if [ "$f" = "bfa_dp_$NK.hip" ]; then OBJS="$OBJS $OUT/${NK}_$v.o"; else OBJS="$OBJS build/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libbfa_${NK}_$v.so $OBJS
done
