cd $GRAFT_REPO_ROOT
for s in $(seq 301 316); do timeout 900 python tests/soak.py 250 $s --record gpurun_out/r5_soak_planner.json 2>&1 | tail -1; done
