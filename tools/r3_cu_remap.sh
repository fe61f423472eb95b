ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { BFA_BENCH_DUMP_K1=1 python bench.py --steps 20 --warmup 5 --no-cpu "$@" 2>&1 | grep "mean per buffer\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('   ms/step %.4f  K1 %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
    else: print('  ', l.strip())"; }
for i in 1 2; do
echo "xcd remap only"; run
for P in 32 16 4 2; do echo "cu remap $P"; BFA_HIP_LIBRARY=$ROOT/bournemouth-forced-aligner_amd/variants/libbfa_cu$P.so run; done
done
