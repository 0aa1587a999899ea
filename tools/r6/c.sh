#!/bin/bash
# Round 6, call c: kernel-clock traces of the C3 step through four release-free queues, barrier bit on every step against chained
cd $GRAFT_REPO_ROOT
O=gpurun_out
export SAFELIFE_HIP_LIB=$PWD/tools/lib_trace.so SAFELIFE_HIP_LIB_ANY_ABI=1
for ch in 0 1; do
  echo "== chained=$ch"
  timeout 300 python tools/trace_overlap.py 12 --queues 4 --fences none --chained $ch 2>&1 | grep -v amdgpu.ids | tail -32
done > $O/r6c_trace.txt 2>&1
echo "== chained=1 spread=1" >> $O/r6c_trace.txt
timeout 300 python tools/trace_overlap.py 12 --queues 4 --fences none --chained 1 --spread 1 2>&1 | grep -v amdgpu.ids | tail -18 >> $O/r6c_trace.txt
echo "== chained=0 spread=1" >> $O/r6c_trace.txt
timeout 300 python tools/trace_overlap.py 12 --queues 4 --fences none --chained 0 --spread 1 2>&1 | grep -v amdgpu.ids | tail -18 >> $O/r6c_trace.txt
cat $O/r6c_trace.txt
