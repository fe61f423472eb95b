#!/bin/bash
# tools/build_variant.sh <name> <unit> <flags...> -- an A/B library: libbfa_hip.so with ONE translation unit recompiled with extra
# -D flags (e.g. `tools/build_variant.sh t_base bfa_dp_nk5_p7 -DBFA_DBG_TIMES` for tools/lone_times.py), written to
# bournemouth-forced-aligner_amd/variants/libbfa_<name>.so; run with BFA_HIP_LIBRARY=<that file>.  Needs an up-to-date `make -C csrc`.
set -e
NAME=$1; UNIT=$2; shift 2
CS=$(dirname "$0")/../bournemouth-forced-aligner_amd/csrc
mkdir -p $CS/../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c $CS/$UNIT.hip -o /tmp/variant_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/../variants/libbfa_$NAME.so $(ls $CS/build/*.o | grep -v "/$UNIT.hip.o") /tmp/variant_$NAME.o
echo "built $CS/../variants/libbfa_$NAME.so"
