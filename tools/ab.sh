#!/bin/bash
# tools/r4_abn.sh "<variant names>" [reps] -- one box, interleaved: the build and variant libraries (tools/build_variant.sh) on
# the mixed-length workloads (one rank's C4 shard, one unsorted call, C4 at N = 1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
ms() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], 'parity', (d.get('parity_sample') or {}).get('mismatching_utterances'))"; }
for rep in $(seq 1 ${2:-2}); do
  for lib in build $1; do
    if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
    python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 64 2>/dev/null | ms "$lib shard"
    python bench.py --ragged --steps 30 2>/dev/null | ms "$lib ragged"
    [ -z "$NO_C4" ] && python bench.py --config c4 --steps 6 --warmup 2 --parity-sample 64 2>/dev/null | ms "$lib c4"
  done
done
