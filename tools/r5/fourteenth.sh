#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "pool_refresh" 2>&1 | tail -5 > $O/r5o_pytest.txt
cat $O/r5o_pytest.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5o_bench_k20.txt 2> $O/r5o_bench_k20.err
tail -c 400 $O/r5o_bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5o_bench_k20.txt').read().strip().splitlines()[-1])
print('K=%d %.3f us/step frac %.3f' % (d['steps'], d['ms_per_step']*1e3, d['roofline']['frac']))
for k,v in sorted(d.get('extra',{}).items()):
    if ("pool_" in k) and not k.endswith("note"): print('  %-60s %s' % (k,v))
PY
