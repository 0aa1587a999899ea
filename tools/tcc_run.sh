#!/bin/bash
# L2 (TCC) counters of the queue-stepped bench kernel in both fence modes, separate rocprofv3 --pmc passes (no tracing):
# hits / misses / requests, the fabric-side read and write requests, write-backs and invalidations.
# usage: tools/tcc_run.sh <tag>      -> gpurun_out/<tag>_tcc_<mode>.txt
set -u
TAG=$1
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for MODE in agent none; do
  OUT=$ROOT/gpurun_out/tcc_${TAG}_$MODE
  mkdir -p $OUT
  RES=$ROOT/gpurun_out/${TAG}_tcc_$MODE.txt
  echo "# bench.py --steps 100 --warmup 10 --queues 4 --queue-fences $MODE --stream-leg 0 (2048-env slice launches); rocprofv3 --pmc, one pass per line group" > $RES
  pass() {
    local name=$1; shift
    rocprofv3 --pmc "$@" -d $OUT/$name -- python $ROOT/bench.py --steps 100 --warmup 10 --cpu-baseline 0 --extras 0 --rollout 0 --queues 4 --queue-fences $MODE --stream-leg 0 > $OUT/$name.log 2>&1
    local db=$(ls $OUT/$name/*/*_results.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep -E "rollout" >> $RES
    grep -h "queue_fences\"" $OUT/$name.log | head -0
    rm -rf $OUT/$name
  }
  pass p1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pass p2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pass p3 TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_ALL_TC_OP_INV_EVICT_sum
  pass p4 TCC_WRITE_sum TCC_NORMAL_EVICT_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
  pass p5 TCC_STREAMING_REQ_sum TCC_BYPASS_REQ_sum TCC_PROBE_sum TCC_ATOMIC_sum
  pass p6 FETCH_SIZE
  pass p7 WRITE_SIZE
  rm -rf $OUT
done
