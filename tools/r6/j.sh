#!/bin/bash
# Round 6, call j: the first steps of a region on the kernels' own clocks (are they slower than the steady state?)
cd $GRAFT_REPO_ROOT
O=gpurun_out
SAFELIFE_HIP_LIB=$PWD/tools/lib_trace.so SAFELIFE_HIP_LIB_ANY_ABI=1 timeout 300 python tools/trace_overlap.py 20 --queues 4 --fences none --stage 1 2>&1 | grep -v amdgpu.ids > $O/r6j_trace20.txt
grep -E "slice 0|slice 3|steady" $O/r6j_trace20.txt
