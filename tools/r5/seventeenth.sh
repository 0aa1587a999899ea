#!/bin/bash
# timing-only builds of the episode-end pass: no draws / no counting / neither (WRONG results by construction)
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
for rep in 1 2; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  for lib in nodraw nocount noboth; do
    echo -n "lib_$lib: "; SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  done
done > gpurun_out/r5r_timing_only_pass.txt 2>&1
cat gpurun_out/r5r_timing_only_pass.txt
