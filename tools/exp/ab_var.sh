#!/bin/bash
# A/B of build variants on one box, alternating with the tree's library:
#   tools/exp/ab_var.sh <variant.so> <c3|c5> [reps]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
V=$1; WHAT=$2; REPS=${3:-3}
for rep in $(seq $REPS); do
  for lib in tree $V; do
    if [ $lib != tree ]; then export SAFELIFE_HIP_LIB=$PWD/$lib SAFELIFE_HIP_LIB_ANY_ABI=1; else unset SAFELIFE_HIP_LIB SAFELIFE_HIP_LIB_ANY_ABI; fi
    if [ $WHAT = c3 ]; then KFIT_STAGE=1 timeout 300 python tools/exp/kfit.py 1 none 5 2>&1 | grep -E "K= 20|K=400|elapsed" | sed "s|^|$lib: |"
    else timeout 300 python tools/exp/c5_steps.py 2>&1 | grep "us/step" | tail -2 | sed "s|^|$lib: |"; fi
  done
done
