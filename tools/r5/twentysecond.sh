#!/bin/bash
# C5 with the pass under the steps: step workgroups per CU capped through their LDS request
cd $GRAFT_REPO_ROOT
{
for pad in 0 41000 55000 70000; do
  SAFELIFE_STEP_LDS_MIN=$pad timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
done
OVERLAP=0 timeout 300 python tools/exp/c5_se.py 2>&1 | grep "us/step"
} > gpurun_out/r5w_c5_overlap.txt 2>&1
cat gpurun_out/r5w_c5_overlap.txt
