#!/bin/bash
# Round 6, call k: the driver's bench command on the tree with pool_ready / staged regions; every extra printed
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/r6k_bench.err | tail -1 > $O/r6k_bench.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6k_bench.txt').read().strip().splitlines()[-1])
r=d['roofline']
print('K=20: %.3f us/step frac %.3f value %.4g' % (d['ms_per_step']*1e3, r['frac'], d['value']))
for k,v in r.items():
    if k.endswith('_us') or k.endswith('_per_s'): print('  roofline.%s = %s' % (k, v))
for k,v in d['extra'].items():
    if isinstance(v,(int,float)): print('  extra.%s = %.4g' % (k, v))
    elif isinstance(v,dict):
        for kk,vv in v.items():
            if isinstance(vv,(int,float)): print('  extra.%s.%s = %.4g' % (k,kk,vv))
print(d['cpu_baseline']['parity_check'])
PY
