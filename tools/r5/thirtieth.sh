#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
LIB=$1
{
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 2400 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
  SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -1
done
timeout 300 python tools/occ_bench.py append_spawn_25 8192 1000 2>&1 | grep -v amdgpu
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/occ_bench.py append_spawn_25 8192 1000 2>&1 | grep -v amdgpu
} > gpurun_out/r5ae_$LIB.txt 2>&1
cat gpurun_out/r5ae_$LIB.txt
