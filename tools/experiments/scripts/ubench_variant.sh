#!/bin/bash
# tools/ubench/variant.sh NAME "<extra hipcc flags>" [files...] -- a copy of the library built with extra flags
# (all sources, or only the listed ones recompiled and the rest taken from csrc/build) -> tools/ubench/dbg/libbfa_NAME.so
set -eu
NAME=$1; EXTRA=$2; shift 2
cd "$(dirname "$0")/../../bournemouth-forced-aligner_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I. -I../../include"
OUT=../../tools/ubench/dbg/$NAME
mkdir -p $OUT
ALL="bfa_kernels.hip bfa_dp_nk5_p3.hip bfa_dp_nk2_p3.hip bfa_dp_nk5_p2.hip bfa_dp_nk2_p2.hip bfa_dp_nk5_p4.hip bfa_dp_nk2_p4.hip bfa_dp_nk5_p5.hip bfa_dp_nk2_p5.hip bfa_dp_nk5_p6.hip bfa_dp_nk2_p6.hip bfa_dp_nk5_p7.hip bfa_dp_nk2_p7.hip bfa_dp_nk2.hip bfa_dp_nk5.hip bfa_dp_nk8.hip bfa_backtrace.hip bfa_segment.hip bfa_post.hip bfa_stitch.hip bfa_capi.cpp"
SEL=${*:-$ALL}
PIDS=""
for f in $SEL; do rm -f $OUT/$f.o; /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f -o $OUT/$f.o & PIDS="$PIDS $!"; done
for p in $PIDS; do wait $p || { echo "variant.sh: a compile failed"; exit 1; }; done
OBJS=""
for f in $ALL; do if [ -f $OUT/$f.o ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS build/$f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/dbg/libbfa_$NAME.so $OBJS
echo built tools/ubench/dbg/libbfa_$NAME.so
