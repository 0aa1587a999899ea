#!/bin/bash
# A/B of two builds on one box, alternating: the tree's library against tools/exp/lib_before.so --
# C3 regions as bench.py times them (kfit, staged, episode ends), C5's stepping (c5_steps), C4's (append-spawn 25x25)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
for rep in 1 2 3; do
  for lib in tree before; do
    if [ $lib = before ]; then export SAFELIFE_HIP_LIB=$PWD/tools/exp/lib_before.so SAFELIFE_HIP_LIB_ANY_ABI=1; else unset SAFELIFE_HIP_LIB SAFELIFE_HIP_LIB_ANY_ABI; fi
    KFIT_STAGE=1 timeout 300 python tools/exp/kfit.py 1 none 5 2>&1 | grep -E "K= 20|K=400|elapsed" | sed "s/^/$lib: /"
    timeout 300 python tools/exp/c5_steps.py 2>&1 | grep "us/step" | tail -2 | sed "s/^/$lib: /"
    [ $rep = 1 ] && timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 2>&1 | grep "us/step" | tail -1 | sed "s/^/$lib: /"
  done
done
