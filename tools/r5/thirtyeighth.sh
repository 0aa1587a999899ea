#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for rep in 1 2; do
  echo -n "spread:    "; timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
  echo -n "no resets: "; NO_SPREAD=1 timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
  echo -n "spread, no goal cache:    "; SAFELIFE_GOAL_CACHE=0 timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
  echo -n "no resets, no goal cache: "; SAFELIFE_GOAL_CACHE=0 NO_SPREAD=1 timeout 300 python tools/exp/c5_steps.py prune_still_25 8192 4 2>&1 | grep "us/step" | tail -1
done
} > gpurun_out/r5al_resets.txt 2>&1
cat gpurun_out/r5al_resets.txt
