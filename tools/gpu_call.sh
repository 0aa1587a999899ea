#!/bin/bash
# One script for the GPU calls of a round (replaces the per-call scripts of rounds 5-6: their results are under profiles/).
#   gpurun --timeout 1800 -- bash tools/gpu_call.sh <tag> <task> [<task> ...]
# Results land in gpurun_out/<tag>_<task>.txt.  Tasks:
#   suite            python -m pytest tests -m gpu -x -q
#   bench20          the driver's command (python bench.py --steps 20 --warmup 5), one line of the numbers that matter
#   bench400         python bench.py --steps 400 --warmup 40 --extras 0 --rollout 0
#   kfit[:spread]    region-length fit, tools/exp/kfit.py <spread> none 5   (fixed cost + per-step cost of a timed region)
#   trace[:args]     kernel-clock timeline on a -DSL_TRACE build (tools/ab_build.sh trace -DSL_TRACE -DSL_DEV_SHAPES first):
#                    tools/trace_overlap.py 12 --queues 4 --fences none <args>
#   prof             rocprofv3 --kernel-trace --stats of the bench + the PMC passes (tools/prof_run.sh; agent fences)
#   soak[:seconds]   tools/soak.py (randomised differential run against the oracle)
#   soakmulti[:s]    tools/soak_multi.py (the multi-agent step against the oracle)
#   abcache          the goal-word cache on / off (SAFELIFE_GOAL_CACHE) at K = 20 and K = 400, alternating
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
O=gpurun_out; mkdir -p $O
TAG=$1; shift
summ() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print('K=%d: %.3f us/step frac %.3f value %.4g | ' % (d['steps'], d['ms_per_step'] * 1e3, r['frac'], d['value'])
      + ' '.join('%s %.2f' % (k[:-3], r[k]) for k in sorted(r) if k.endswith('_us') and r[k])
      + ' | parity %s' % (d.get('cpu_baseline') or {}).get('parity_check', {}).get('bit_exact'))
PY
}
for task in "$@"; do
  name=${task%%:*}; arg=${task#*:}; [ "$arg" = "$task" ] && arg=""
  out=$O/${TAG}_${name}.txt
  case $name in
    suite)    ( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $out ;;
    bench20)  timeout 900 python bench.py --steps 20 --warmup 5 2>$O/${TAG}_bench20.err | tail -1 > $O/${TAG}_bench20.json; summ $O/${TAG}_bench20.json > $out ;;
    bench400) timeout 600 python bench.py --steps 400 --warmup 40 --extras 0 --rollout 0 2>/dev/null | tail -1 > $O/${TAG}_bench400.json; summ $O/${TAG}_bench400.json > $out ;;
    kfit)     timeout 300 python tools/exp/kfit.py ${arg:-1} none 5 2>&1 | grep -v amdgpu.ids > $out ;;
    trace)    SAFELIFE_HIP_LIB=$PWD/tools/lib_trace.so SAFELIFE_HIP_LIB_ANY_ABI=1 timeout 300 python tools/trace_overlap.py 12 --queues 4 --fences none $arg 2>&1 | grep -v amdgpu.ids > $out ;;
    prof)     bash tools/prof_run.sh $TAG > /dev/null 2>&1; head -4 $O/${TAG}_kernel_trace.txt > $out; grep -E "FETCH_SIZE|WRITE_SIZE" $O/${TAG}_pmc.txt >> $out ;;
    soak)     ( timeout 1200 python tools/soak.py ${arg:-120} 2>&1 | tail -5 ) > $out ;;
    soakmulti) ( timeout 1200 python tools/soak_multi.py ${arg:-120} 2>&1 | tail -3 ) > $out ;;
    abcache)  # the goal-word cache on / off at the driver's K = 20 and at K = 400, alternating, same box
              for rep in 1 2 3; do for gc in 1 0; do for k in 20 400; do
                w=$([ $k = 400 ] && echo 40 || echo 5)
                SAFELIFE_GOAL_CACHE=$gc timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 --stream-leg 0 2>/dev/null | tail -1 > $O/.ab.json
                echo "goal cache $gc: $(summ $O/.ab.json)"
              done; done; done > $out ;;
    *)        echo "unknown task $task" ;;
  esac
  echo "== $task"; cat $out
done
