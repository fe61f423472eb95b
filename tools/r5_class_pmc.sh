cd $GRAFT_REPO_ROOT
for cfg in "1000 40 4096" "2000 80 2048" "2900 116 1408"; do set -- $cfg
  bash tools/pmc.sh cls_$1 python $PWD/bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu --no-confidences --no-uniform-hint --frames $1 --tokens $2 --batch $3 --settle-ms 0 --min-timed-steps 6 --kernel-leg-steps 20 > /dev/null 2>&1
  echo "== T=$1 S=$2 B=$3 frames=$(( $1 * $3 ))"; grep -A24 "^k_mix" gpurun_out/pmc_cls_$1/summary.txt | grep -E "k_mix|INSTS_VALU|INSTS_SALU|INSTS_LDS|GRBM|WAVE_CYCLES|WAIT_INST_ANY|ACTIVE_INST_VALU"
done
