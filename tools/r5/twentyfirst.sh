#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5v_bench_k20.txt 2> $O/r5v_bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5v_bench_k20.txt').read().strip().splitlines()[-1])
print('K=%d %.3f us/step frac %.3f parity %s' % (d['steps'], d['ms_per_step']*1e3, d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_check')))
for k,v in sorted(d.get('extra',{}).items()):
    if not k.endswith("note") and isinstance(v,(int,float,str)): print('  %-60s %s' % (k,v))
PY
