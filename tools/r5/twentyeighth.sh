#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
SAFELIFE_HIP_LIB=$E/lib_regs1.so timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -2
for rep in 1 2 3; do
for lib in regs0 regs1 dp4; do
  echo -n "$lib: "; SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
done
done
} > gpurun_out/r5ac_regs.txt 2>&1
cat gpurun_out/r5ac_regs.txt
