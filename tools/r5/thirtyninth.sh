#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
LIB=$1
{
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not 26 and not 20 and not 15 and not 10 and not 8 and not 12 and not 16 and not 24 and not 30 and not 32 and not 40 and not 48" 2>&1 | tail -3
for rep in 1 2 3; do
  echo -n "tree spread: "; timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
  echo -n "$LIB spread: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
done
echo -n "tree C4 spread: "; timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
echo -n "$LIB C4 spread: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
echo -n "tree C5 spread: "; timeout 300 python tools/exp/c5_steps.py navigation_64 4096 4 2>&1 | grep "us/step" | tail -1
echo -n "$LIB C5 spread: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/c5_steps.py navigation_64 4096 4 2>&1 | grep "us/step" | tail -1
} > gpurun_out/r5am_$LIB.txt 2>&1
cat gpurun_out/r5am_$LIB.txt
