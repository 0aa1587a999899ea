#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
for rep in 1 2; do
for lib in j0 j1; do
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py navigation_64 4096 4 2>&1 | grep "us/step" | tail -1
done
done
} > gpurun_out/r5ak_jump1.txt 2>&1
cat gpurun_out/r5ak_jump1.txt
