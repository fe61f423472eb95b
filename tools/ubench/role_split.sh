#!/bin/bash
# Role-time experiment for K1 (producer = softmax wave, consumer = DP wave): builds two extra copies of the
# library with one role stubbed out (results are wrong, only the kernel time is of interest; the sliding-window
# kernels end early when a stubbed role leaves every window state at the sentinel, so time the full-layout kernels:
# bench.py --no-window).  Run here (cross-compile), then time on the GPU box with BFA_HIP_LIBRARY=... bench.py.
set -eu
NK=${1:-nk5}   # nk5: the 67-class head, nk2: the 17-class head
HERE="$(dirname "$0")"
for v in NOPRODUCE NOCONSUME; do
  bash "$HERE/variant.sh" ${NK}_$v "-DBFA_DBG_$v" bfa_dp_${NK}_p2.hip bfa_dp_${NK}_p3.hip bfa_dp_${NK}_p4.hip bfa_dp_${NK}_p5.hip
done
