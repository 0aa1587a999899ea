#!/bin/bash
# Round 5, second GPU call: the goal-word cache -- suite, soak, and same-box A/B of the bench with the cache off / on.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r5b_pytest.txt
( timeout 200 python tools/soak.py 120 5 2>&1 | tail -5 ) > $O/r5b_soak.txt
for rep in 1 2 3; do
  for gc in 0 1; do
    for k in 400 20; do
      w=$([ $k = 400 ] && echo 40 || echo 5)
      SAFELIFE_GOAL_CACHE=$gc timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cache=$gc K=%d %.3f us/step frac %.3f host %.2f us fences %s' % (d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3, d['config']['queue_fences'][:5]))"
    done
  done
done > $O/r5b_ab.txt 2>&1
# one full default line with the replay (parity of the timed run) and the extras
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r5b_bench_k20_full.txt 2>&1
cat $O/r5b_pytest.txt $O/r5b_soak.txt $O/r5b_ab.txt
