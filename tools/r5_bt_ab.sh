# tools/r5_bt_ab.sh -- workgroups of k_backtrace (BFA_BT_GRID; a resident wave holds 128 VGPRs: 4096 of them are the whole register
# file) and of k_postconf (BFA_POST_GRID) on the headline step with four batches in flight; one box, interleaved
cd $GRAFT_REPO_ROOT
last() { grep "^{" | tail -1; }
run() { name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | last | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name headline %.4f ms  alignment_only %.4f  K1 %.4f' % (d['ms_per_step'], d['alignment_only']['ms_per_step'], d['roofline']['kernel_ms']))"
}
for rep in 1 2 3; do
  run "bt=all post=all" BFA_X=0
  run "bt=2048 post=all" BFA_BT_GRID=2048
  run "bt=1024 post=all" BFA_BT_GRID=1024
  run "bt=512 post=all" BFA_BT_GRID=512
  run "bt=1024 post=1024" BFA_BT_GRID=1024 BFA_POST_GRID=1024
  run "bt=2048 post=2048" BFA_BT_GRID=2048 BFA_POST_GRID=2048
done
