#!/bin/bash
# Slices on the library's AQL queues against HIP-stream slices on one box (K = 400 and K = 20, no CPU leg).
# usage (on the GPU box): tools/queues_sweep.sh [bench args] > gpurun_out/queues_sweep.txt
for rep in 1 2; do
for cfg in "streams 2" "queues 2" "queues 3" "queues 4" "queues 6" "queues 8"; do
  set -- $cfg
  for k in 400 20; do
    w=$([ $k = 400 ] && echo 40 || echo 5)
    if [ $1 = streams ]; then a="--queues 0 --slices $2"; else a="--queues $2"; fi
    line=$(timeout 200 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 $a $EXTRA 2>/dev/null | tail -1)
    python3 -c "
import json,sys
try:
    d=json.loads(sys.argv[1]); r=d['roofline']
    print('%-8s %s  K=%-3d %7.3f us/step  frac %.3f  host %.2f us  [%s]' % (sys.argv[2], sys.argv[3], d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3, d['config'].get('stepping')))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e, sys.argv[1][:200])" "$line" $1 $2
  done
done
done
