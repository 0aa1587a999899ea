#!/bin/bash
# premise check: the episode-end pass at reduced occupancy (LDS padding), alone
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
mkdir -p gpurun_out
for pad in 0 12000 23000 36000 70000; do
  echo -n "SL_OCC_LDS_PAD=$pad: "
  SL_OCC_LDS_PAD=$pad SAFELIFE_HIP_LIB=$E/lib_occpad.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
done > gpurun_out/r5h_pass_vs_occupancy.txt 2>&1
cat gpurun_out/r5h_pass_vs_occupancy.txt
