#!/bin/bash
# Kernel trace + PMC passes of the bench command on the GPU box; summaries land in gpurun_out/<tag>_*.txt
# usage: tools/prof_run.sh <tag> [extra bench args...]
#   kernel trace : rocprofv3 --kernel-trace --stats of  bench.py --steps 400 --warmup 40 --queues 4 --queue-fences none
#                  (the headline's launcher; rocprofv3 serialises dispatches across queues: per-launch durations are what
#                  to read, the queues' overlap is in the kernels' own clocks, tools/trace_overlap.py)
#   PMC passes   : tools/pmc_run.sh with --queue-fences agent (per-dispatch counters cannot attribute release-free steps)
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $ROOT/bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0 --queues 4 --queue-fences none --stream-leg 0 "$@" > $OUT/kt.log 2>&1
db=$(ls $OUT/kt/*/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db | head -12 > $ROOT/gpurun_out/${TAG}_kernel_trace.txt
[ -n "$db" ] && python $ROOT/tools/timeline.py $db >> $ROOT/gpurun_out/${TAG}_kernel_trace.txt 2>&1
grep '^{' $OUT/kt.log | tail -1 >> $ROOT/gpurun_out/${TAG}_kernel_trace.txt
rm -rf $OUT/kt
cd $ROOT && bash tools/pmc_run.sh $TAG --queues 4 --queue-fences agent --stream-leg 0 "$@" > $ROOT/gpurun_out/${TAG}_pmc.txt 2>&1
rm -rf $ROOT/gpurun_out/pmc_$TAG/*/ 2>/dev/null
