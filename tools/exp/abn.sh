#!/bin/bash
for i in 1 2; do
  for lib in "" "$@"; do
    SAFELIFE_HIP_LIB=$lib python bench.py --steps 400 --warmup 40 --rollout 0 --extras 0 --cpu-baseline 0 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-current}', round(d['ms_per_step']*1000,2), 'us/step')"
  done
done
