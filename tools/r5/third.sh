#!/bin/bash
# Round 5, third GPU call: where the goal-word cache's time went (kernel-clock phases, cache off / on), leader-wave
# priority A/B, instruction counters with the cache on.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
E=$PWD/tools/exp
for gc in 0 1; do
  echo "== SAFELIFE_GOAL_CACHE=$gc" 
  SAFELIFE_GOAL_CACHE=$gc SAFELIFE_HIP_LIB=$E/lib_trace.so timeout 300 python tools/trace_overlap.py 10 --queues 4 --fences none 2>&1 | tail -16
done > $O/r5c_trace_cache.txt 2>&1
for rep in 1 2 3; do
  for lib in base prio1 prio3; do
    for k in 400 20; do
      w=$([ $k = 400 ] && echo 40 || echo 5)
      SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib K=%d %.3f us/step frac %.3f host %.2f us' % (d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3))"
    done
  done
done > $O/r5c_ab_prio.txt 2>&1
export SAFELIFE_HIP_LIB=$E/lib_base.so
bash tools/pmc_any.sh r5c_c3_cache "rollout" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
  -- python bench.py --steps 100 --warmup 10 --cpu-baseline 0 --extras 0 --rollout 0 --stream-leg 0 > /dev/null 2>&1
unset SAFELIFE_HIP_LIB
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "goal_word_cache or full_size_env or queue_stepping_vs" 2>&1 | tail -5 ) > $O/r5c_pytest.txt
( timeout 200 python tools/soak.py 100 7 2>&1 | tail -4 ) > $O/r5c_soak.txt
cat $O/r5c_trace_cache.txt $O/r5c_ab_prio.txt $O/r5c_pytest.txt $O/r5c_soak.txt
