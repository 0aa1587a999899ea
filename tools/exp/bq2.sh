#!/bin/bash
label=$1; shift
timeout 600 python bench.py "$@" 2>/dev/null | grep -E "^\{" | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; x=d.get('extra',{})
print('$label', 'wall %.2f dev %.2f host %.2f' % (d['ms_per_step']*1e3, r['launch_ms']*1e3, r['host_enqueue_ms_per_step']*1e3), 'c5 %.1f c5se %.1f' % (x.get('c5_navigation_64_us_per_step',0), x.get('c5_with_side_effects_us_per_step',0)))"
