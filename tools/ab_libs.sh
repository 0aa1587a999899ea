#!/bin/bash
# A/B of two builds of the library on ONE box: alternating runs of the stream path, K = 400 and K = 20.
# usage: tools/ab_libs.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    for k in 400 20; do
      w=$([ $k = 400 ] && echo 40 || echo 5)
      line=$(SAFELIFE_HIP_LIB_ANY_ABI=1 SAFELIFE_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 "$@" 2>/dev/null | tail -1)
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']
print('%-40s K=%-3d %7.3f us/step  device %.3f us  host %.2f us' % (sys.argv[2], d['steps'], d['ms_per_step']*1e3, r.get('stream_leg_launch_ms', r.get('device_launch_ms', 0))*1e3, r['host_enqueue_ms_per_step']*1e3))" "$line" "$lib"
    done
  done
done
