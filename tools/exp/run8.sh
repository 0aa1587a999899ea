#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sl in 1 2; do
rocprofv3 --kernel-trace -d $OUT/kt$sl -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --cpu-baseline 0 --extras 0 --rollout 0 --slices $sl > $OUT/kt$sl.log 2>&1
db=$(ls $OUT/kt$sl/*/*_results.db | head -1)
python $GRAFT_REPO_ROOT/tools/timeline.py $db > $GRAFT_REPO_ROOT/gpurun_out/timeline_slices$sl.txt 2>&1
done
rm -rf $OUT
cat $GRAFT_REPO_ROOT/gpurun_out/timeline_slices2.txt | tail -40
tail -4 $GRAFT_REPO_ROOT/gpurun_out/timeline_slices1.txt
