import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], 'step_ms=%.4f'%d['ms_per_step'], 'k1_ms=%.4f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'])
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-300:])
