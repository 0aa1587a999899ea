#!/bin/bash
# forced one-rank exchange (RCCL to itself) against no exchange, by queue count and region length: us per step
for q in 4 3 2; do for k in 20 400; do for g in 0 1; do
  w=$([ $k = 400 ] && echo 40 || echo 5)
  for rep in 1 2; do
  line=$(SAFELIFE_FORCE_GATHER=$g timeout 200 python bench.py --queues $q --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 "$@" 2>/dev/null | tail -1)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']; c=d['config']
print('queues %s K=%-3d gather %s  %7.3f us/step  host %.2f us  windows %s every %s exposed %.1f us' % (sys.argv[2], d['steps'], sys.argv[3], d['ms_per_step']*1e3, r['host_enqueue_ms_per_step']*1e3, d.get('gather_windows_in_region', c.get('gather_windows_in_region')), d.get('gather_every', c.get('gather_every')), 1e3*(d.get('gather_exposed_ms') or c.get('gather_exposed_ms') or 0)))" "$line" $q $g
  done
done; done; done
