ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_clk; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in normal same; do
  if [ $mode = same ]; then export BFA_BENCH_SAME_INPUT=1; else unset BFA_BENCH_SAME_INPUT; fi
  rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/$mode -o p -- python $ROOT/bench.py --steps 100 --warmup 20 --no-cpu > $OUT/$mode.log 2>&1
done
python - <<PY
import csv,glob,collections
for mode in ("normal","same"):
    f=glob.glob("$OUT/%s/*counter_collection.csv"%mode)[0]
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f)):
        if "k_dp4w" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
    d=sum(dur[20:])/len(dur[20:])
    print(mode, "dur_us=%.1f"%d, {k: round(sum(v[20:])/len(v[20:])) for k,v in agg.items()}, "clk_GHz(GRBM/8/dur)=%.3f"%(sum(agg["GRBM_GUI_ACTIVE"][20:])/len(dur[20:])/8/d/1e3))
PY
