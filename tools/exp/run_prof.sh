#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_run.sh round2_a_slices1 --slices 1
bash tools/prof_run.sh round2_a_slices2 --slices 2
SAFELIFE_HIP_LIB=tools/lib_trace.so SL_TRACE_ENVS=4096 python tools/trace_phases.py > gpurun_out/round2_a_phase_trace_4096.txt 2>&1
SAFELIFE_HIP_LIB=tools/lib_trace.so SL_TRACE_ENVS=8192 python tools/trace_phases.py > gpurun_out/round2_a_phase_trace_8192.txt 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --extras 0 --rollout 0 | tail -1 | cut -c1-1200; done > gpurun_out/round2_a_bench_k20.txt 2>&1
python bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0 | tail -1 > gpurun_out/round2_a_bench_k400.txt 2>&1
ls -la gpurun_out/ | tail -12
