#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "lib_$lib: "; SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  done
done > gpurun_out/r5t_pass.txt 2>&1
cat gpurun_out/r5t_pass.txt
