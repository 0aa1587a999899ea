#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
for rep in 1 2; do
for lib in dp0 dp1 dp2; do
  echo -n "$lib: "; SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
done
done
} > gpurun_out/r5ab_deal.txt 2>&1
cat gpurun_out/r5ab_deal.txt
