#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for cfg in "4096 4" "3072 3" "4096 3" "4096 2" "2048 2" "3072 4" "6144 4" "8192 4"; do
  timeout 300 python tools/exp/c5_steps.py navigation_64 $cfg 2>&1 | grep "us/step" | tail -1
done
} > gpurun_out/r5z_c5_slices.txt 2>&1
cat gpurun_out/r5z_c5_slices.txt
