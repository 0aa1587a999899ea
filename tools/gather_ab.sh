#!/bin/bash
# tools/gather_ab.sh libA libB ...: forced one-rank exchange against none, K = 20 / 400, per library ("-" = in-tree)
for rep in 1 2; do for lib in "$@"; do for k in 20 400; do for g in 0 1; do
  w=$([ $k = 400 ] && echo 40 || echo 5)
  L=$([ "$lib" = "-" ] && echo "" || echo "$PWD/$lib")
  line=$(SAFELIFE_HIP_LIB_ANY_ABI=1 SAFELIFE_HIP_LIB=$L SAFELIFE_FORCE_GATHER=$g timeout 200 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/tmp/gab.err | tail -1)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']; c=d['config']
print('%-20s K=%-3d gather %s  %7.3f us/step  host %.2f us  fences %s' % (sys.argv[2], d['steps'], sys.argv[3], d['ms_per_step']*1e3, r['host_enqueue_ms_per_step']*1e3, str(c.get('queue_fences'))[:5]))" "$line" "$lib" $g
  grep -i "bench:" /tmp/gab.err | head -2
done; done; done; done
