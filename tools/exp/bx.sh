#!/bin/bash
# extras of one bench run: bx.sh <label> [bench args]
label=$1; shift
timeout 600 python bench.py --steps 20 --warmup 5 --rollout 0 --cpu-baseline 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); x=d['extra']
print('$label', 'c3 %.2f' % (d['roofline']['launch_ms']*1e3), ' '.join('%s %.1f' % (k.replace('_us_per_step',''), v) for k,v in x.items() if k.endswith('us_per_step')))"
