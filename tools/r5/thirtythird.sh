#!/bin/bash
cd $GRAFT_REPO_ROOT
SL_AQL_TIMELINE=1 timeout 300 python bench.py --steps 20 --warmup 5 --extras 0 --rollout 0 --cpu-baseline 0 --stream-leg 0 > gpurun_out/r5ag_timeline.out 2> gpurun_out/r5ag_timeline.err
grep -n "aql timeline" gpurun_out/r5ag_timeline.err | tail -3
awk '/aql timeline/{n++} {if(n>0) print n": "$0}' gpurun_out/r5ag_timeline.err | tail -120 > gpurun_out/r5ag_timeline.txt
tail -100 gpurun_out/r5ag_timeline.txt
