#!/bin/bash
# tree = round-by-round deal + transposed flush; dev lib on top of it as $1
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
LIB=$1
{
echo "== parity, lib_$LIB"
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -3
for rep in 1 2; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "lib_$LIB: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
done
echo "== full GPU suite, tree"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
} > gpurun_out/r5u_$LIB.txt 2>&1
cat gpurun_out/r5u_$LIB.txt
