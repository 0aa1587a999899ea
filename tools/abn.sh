#!/bin/bash
# A/B of several builds of the library on ONE box, default launcher of bench.py: tools/abn.sh "<bench args>" libA.so libB.so ...
# ("-" = the in-tree build)
ARGS=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    for k in 400 20; do
      w=$([ $k = 400 ] && echo 40 || echo 5)
      L=$([ "$lib" = "-" ] && echo "" || echo "$PWD/$lib")
      line=$(SAFELIFE_HIP_LIB_ANY_ABI=1 SAFELIFE_HIP_LIB=$L timeout 200 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 $ARGS 2>/dev/null | tail -1)
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']
print('%-28s K=%-3d %7.3f us/step  device %.3f us  host %.2f us' % (sys.argv[2], d['steps'], d['ms_per_step']*1e3, r.get('stream_leg_launch_ms', r.get('device_launch_ms', 0))*1e3, r['host_enqueue_ms_per_step']*1e3))" "$line" "$lib"
    done
  done
done
