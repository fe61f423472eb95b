ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/variants; rm -f $ROOT/gpurun_out/variants/*.json
for B in 1024 4096; do
for n in hip noproduce noconsume nostore nobandupd noslide nofetch nos4; do
  if [ $n = hip ]; then lib=$ROOT/bournemouth-forced-aligner_amd/libbfa_hip.so; else lib=$ROOT/tools/ubench/dbg/libbfa_$n.so; fi
  BFA_HIP_LIBRARY=$lib python $ROOT/bench.py --no-cpu --batch $B --steps 60 --warmup 20 > $ROOT/gpurun_out/variants/B${B}_$n.json 2>/dev/null
done; done
python $ROOT/tools/ubench/extract.py $ROOT/gpurun_out/variants/*.json
