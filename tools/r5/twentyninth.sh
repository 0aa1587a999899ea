#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
LIB=$1
{
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -2
for rep in 1 2 3; do
  echo -n "tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "$LIB: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
done
} > gpurun_out/r5ad_$LIB.txt 2>&1
cat gpurun_out/r5ad_$LIB.txt
