cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_golden.py -q -m gpu -k "narrow or softmax" 2>&1 | tail -3
j() { grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.4f' % d['ms_per_step'])"; }
python bench.py --config realtext --inflight 1 --hw-queues 4 --no-cpu --parity-sample 0 2>/dev/null | j "rt q4 if1"
python bench.py --config realtext --inflight 3 --hw-queues 4 --no-cpu --parity-sample 0 2>/dev/null | j "rt q4 if3"
python bench.py --config realtext --inflight 1 --hw-queues 8 --no-cpu --parity-sample 0 2>/dev/null | j "rt q8 if1"
python bench.py --config realtext --inflight 3 --hw-queues 8 --no-cpu --parity-sample 0 2>/dev/null | j "rt q8 if3"
bash tools/timeline.sh rt_q4 2 python $GRAFT_REPO_ROOT/bench.py --config realtext --inflight 1 --hw-queues 4 --steps 4 --warmup 2 --no-cpu --parity-sample 0 | grep -v "at::native\|rocprim" | tail -24
