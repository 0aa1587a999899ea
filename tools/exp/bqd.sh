#!/bin/bash
# quick bench line under torch.distributed.run (1 rank): bqd.sh <label> <bench args...>
label=$1; shift
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --extras 0 --rollout 0 --cpu-baseline 0 "$@" 2>&1 | grep -E "^\{|Error|error|Traceback" | tail -1 | python -c "
import sys,json
l=sys.stdin.readlines()[-1]
try:
    d=json.loads(l); r=d['roofline']; print('$label', 'K', d['steps'], 'wall %.2f us' % (d['ms_per_step']*1e3), 'dev %.2f' % (r['launch_ms']*1e3), 'host %.2f' % (r['host_enqueue_ms_per_step']*1e3), 'frac_wall', r.get('frac_wall'))
except Exception: print('$label', l)"
