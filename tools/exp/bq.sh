#!/bin/bash
# quick bench lines: bq.sh <label> <bench args...>
label=$1; shift
python bench.py --extras 0 --rollout 0 --cpu-baseline 0 "$@" 2>&1 | grep -E "^\{" | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$label', 'K', d['steps'], 'wall %.2f us' % (d['ms_per_step']*1e3), 'dev %.2f' % (r['launch_ms']*1e3*r.get('launches_per_step',1) if False else r['launch_ms']*1e3), 'host %.2f' % (r['host_enqueue_ms_per_step']*1e3), 'frac_wall', r.get('frac_wall'))"
