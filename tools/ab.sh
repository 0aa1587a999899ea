#!/bin/bash
# A/B two builds of libsafelife_hip.so in one GPU session, interleaved: tools/ab.sh <other.so> [bench args]
OTHER=$1; shift
for i in 1 2 3; do
  for lib in "" "$OTHER"; do
    SAFELIFE_HIP_LIB=$lib python bench.py --steps 400 --warmup 40 --rollout 32 --cpu-baseline 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-current}', round(d['ms_per_step']*1000,2), 'us/step; rollout', round(8192*32/d['extra']['rollout_env_steps_per_s_per_gpu']*1e6/32,2), 'us/step')"
  done
done
