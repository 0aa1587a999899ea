#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5n_bench_k20.txt 2> $O/r5n_bench_k20.err
tail -c 400 $O/r5n_bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5n_bench_k20.txt').read().strip().splitlines()[-1])
print('K=%d %.3f us/step frac %.3f parity %s' % (d['steps'], d['ms_per_step']*1e3, d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_check')))
for k,v in sorted(d.get('extra',{}).items()):
    if ("c5_with" in k or "pool_" in k) and not k.endswith("note"): print('  %-60s %s' % (k,v))
print(json.dumps(d['roofline'].get('issue_side'))[:300])
PY
