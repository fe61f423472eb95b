#!/bin/bash
# tools/experiments/scripts/r4_lm.sh -- k_one (small uniform calls) with 1 / 2 / 3 producer waves against the two-wave kernel
# (variants/libbfa_old.so), same box, interleaved: config 2 and one utterance per call; then the parity tests and a soak slice
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r4lm
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'], 'alignment only', d.get('alignment_only',{}).get('ms_per_step'))"; }
for rep in 1 2 3; do for lib in build old np3 np1; do
  if [ $lib = build ]; then unset BFA_HIP_LIBRARY; else export BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_$lib.so; fi
  python bench.py --config c2 --steps 50 --warmup 10 --no-cpu --inflight 1 2>/dev/null | j "$lib c2"
  python tools/latency_device.py 2>/dev/null | grep "B=" | sed "s/^/$lib /"
done; done
unset BFA_HIP_LIBRARY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_xwin.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tests/soak.py 150 77 2>&1 | tail -1
