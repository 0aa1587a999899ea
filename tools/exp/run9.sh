#!/bin/bash
cd $GRAFT_REPO_ROOT
SAFELIFE_HIP_LIB=tools/lib_trace.so SL_TRACE_ENVS=4096 python tools/trace_phases.py > gpurun_out/phase_4096.txt 2>&1
SAFELIFE_HIP_LIB=tools/lib_trace.so SL_TRACE_ENVS=8192 python tools/trace_phases.py > gpurun_out/phase_8192.txt 2>&1
cat gpurun_out/phase_4096.txt gpurun_out/phase_8192.txt | grep -v amdgpu.ids
