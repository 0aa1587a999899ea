#!/bin/bash
# Kernel trace + PMC passes of the default bench command on the GPU box; summaries land in gpurun_out/<tag>_*.txt
# usage: tools/prof_run.sh <tag> [bench args...]
# (rocprofv3's kernel tracing serialises dispatches across queues: with --slices 2 the two 4096-env launches of a
#  step show up back to back in the trace instead of overlapped; their per-launch durations are what to read.)
set -u
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $ROOT/bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0 "$@" > $OUT/kt.log 2>&1
db=$(ls $OUT/kt/*/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db | head -12 > $ROOT/gpurun_out/${TAG}_kernel_trace.txt
[ -n "$db" ] && python $ROOT/tools/timeline.py $db >> $ROOT/gpurun_out/${TAG}_kernel_trace.txt 2>&1
grep '^{' $OUT/kt.log | tail -1 >> $ROOT/gpurun_out/${TAG}_kernel_trace.txt
rm -rf $OUT/kt
cd $ROOT && bash tools/pmc_run.sh $TAG "$@" > $ROOT/gpurun_out/${TAG}_pmc.txt 2>&1
rm -rf $ROOT/gpurun_out/pmc_$TAG/*/ 2>/dev/null
