#!/bin/bash
# round-by-round deal of a board's draws (dev lib): parity, then the pass, life_occupancy and C5's step
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
LIB=${1:-stride}
{
echo "== parity, lib_$LIB"
SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect or advance_board or navigation or golden or trace" 2>&1 | tail -3
for rep in 1 2; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "lib_$LIB: "; SAFELIFE_HIP_LIB=$E/lib_$LIB.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
done
} > gpurun_out/r5s_$LIB.txt 2>&1
cat gpurun_out/r5s_$LIB.txt
