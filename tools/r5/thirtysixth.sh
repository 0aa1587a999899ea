#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
for lib in cur skip1 skip2 skip4 skip7; do
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
done
} > gpurun_out/r5aj_c4_timing_only.txt 2>&1
cat gpurun_out/r5aj_c4_timing_only.txt
