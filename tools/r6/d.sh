#!/bin/bash
# Round 6, call d: does a chained dispatch overlap its predecessor when its header carries no acquire fence? (timing experiment)
cd $GRAFT_REPO_ROOT
O=gpurun_out
export SAFELIFE_HIP_LIB=$PWD/tools/lib_trace.so SAFELIFE_HIP_LIB_ANY_ABI=1
for acq in 0 1; do
  echo "== chained=1 SL_AQL_STEP_ACQUIRE=$acq"
  SL_AQL_STEP_ACQUIRE=$acq timeout 300 python tools/trace_overlap.py 12 --queues 4 --fences none --chained 1 2>&1 | grep -v amdgpu.ids | tail -24
done > $O/r6d_trace.txt 2>&1
unset SAFELIFE_HIP_LIB
for acq in 0 1 2; do
  echo "== tree, chained=1 SL_AQL_STEP_ACQUIRE=$acq" >> $O/r6d_trace.txt
  SL_AQL_STEP_ACQUIRE=$acq timeout 300 python tools/exp/kfit.py 0 none 5 1 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/r6d_trace.txt
done
cat $O/r6d_trace.txt
