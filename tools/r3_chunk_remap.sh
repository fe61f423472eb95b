#!/bin/bash
# tools/r3_chunk_remap.sh -- XCD chunk size of the workgroup -> utterance mapping (variants g0 / g4 / g64 / g256 of the ph66 K1 units
# against the build, G = 16): headline K1 per resident batch, mixed-length calls
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
j() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'K1', (d.get('roofline') or {}).get('kernel_ms_per_buffer'), d.get('status_ok', ''))"; }
for i in 1 2; do for v in build; do
  if [ $v = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | j "headline $v"
done; done
for v in build; do
  if [ $v = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_$v.so; fi
  python bench.py --ragged --steps 30 2>/dev/null | j "ragged $v"
  python bench.py --config c4 --global-batch 4096 --steps 20 --warmup 3 --parity-sample 0 2>/dev/null | j "c4 shard $v"
  python bench.py --config c4 --steps 8 --warmup 2 --parity-sample 0 2>/dev/null | j "c4 full $v"
done
