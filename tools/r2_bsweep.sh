ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/bsweep
for lib in bournemouth-forced-aligner_amd/libbfa_hip.so tools/ubench/dbg/libbfa_nos4.so; do
for B in 512 1024 2048 3072 4096 6144 8192; do
  n=$(basename $lib .so)
  BFA_HIP_LIBRARY=$ROOT/$lib python $ROOT/bench.py --no-cpu --batch $B --steps 60 --warmup 20 > $ROOT/gpurun_out/bsweep/${n}_B$B.json 2>/dev/null
done; done
python $ROOT/tools/ubench/extract.py $ROOT/gpurun_out/bsweep/*.json
