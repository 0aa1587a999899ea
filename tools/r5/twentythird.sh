#!/bin/bash
# C5 with the pass launched BEHIND the first steps of the next window (deferred launch), step workgroups capped, step
# queues at high priority
cd $GRAFT_REPO_ROOT
{
DEFER=8 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
DEFER=8 SAFELIFE_STEP_LDS_MIN=55000 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
DEFER=8 SL_AQL_PRIORITY=1 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
DEFER=8 SL_AQL_PRIORITY=1 SAFELIFE_STEP_LDS_MIN=55000 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
DEFER=8 SL_AQL_PRIORITY=1 SAFELIFE_STEP_LDS_MIN=41000 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
} > gpurun_out/r5x_c5_defer.txt 2>&1
cat gpurun_out/r5x_c5_defer.txt
