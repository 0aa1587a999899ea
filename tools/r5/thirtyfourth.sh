#!/bin/bash
# longer randomised differential runs of the final tree against the oracle, fresh seeds
cd $GRAFT_REPO_ROOT
{
for seed in 11 12 13; do
  timeout 900 python tools/soak.py 240 $seed 2>&1 | grep "soak"
  timeout 900 python tools/soak_prims.py 240 $seed 2>&1 | grep "soak"
done
} > gpurun_out/r5ah_soak.txt 2>&1
cat gpurun_out/r5ah_soak.txt
