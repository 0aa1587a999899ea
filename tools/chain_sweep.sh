#!/bin/bash
# A/B runs of the AQL chain's knobs against the HIP-stream path on one box (K = 400 and K = 20, no CPU leg).
# usage (on the GPU box): tools/chain_sweep.sh > gpurun_out/chain_sweep.txt
run() {   # label, env assignments..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    for k in 400 20; do
        local w=$([ $k = 400 ] && echo 40 || echo 5)
        line=$(env "${envs[@]}" timeout 200 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 "$@" 2>/dev/null | tail -1)
        python3 - "$label" "$k" "$line" <<'PY'
import json, sys
label, k, line = sys.argv[1:4]
try:
    d = json.loads(line)
    r = d["roofline"]
    print("%-34s K=%-3s %7.3f us/step  frac %.3f  host %.2f us  [%s]" % (label, k, d["ms_per_step"] * 1e3, r["frac"],
          r["host_enqueue_ms_per_step"] * 1e3, d["config"].get("stepping")))
except Exception as e:
    print("%-34s K=%-3s FAILED %s %r" % (label, k, e, line[:200]))
PY
    done
}
run "streams x2" -- --chain off
run "chain q1" -- --chain on
run "chain q2" -- --chain on --chain-slices 2
run "chain q4" -- --chain on --chain-slices 4
run "chain q8" -- --chain on --chain-slices 8
run "chain q2 acq=none" SL_AQL_ACQUIRE=0 -- --chain on --chain-slices 2
run "chain q4 acq=none" SL_AQL_ACQUIRE=0 -- --chain on --chain-slices 4
run "chain q4 noreadback" SL_AQL_READBACK=0 -- --chain on --chain-slices 4
run "chain q1 barrier" SL_AQL_BARRIER=1 -- --chain on
