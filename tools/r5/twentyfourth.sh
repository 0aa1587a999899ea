#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
for lib in cur skip1 skip2 skip3 skip4 skip7; do
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py 2>&1 | grep "us/step" | tail -2
done
} > gpurun_out/r5y_c5_timing_only.txt 2>&1
cat gpurun_out/r5y_c5_timing_only.txt
