#!/bin/bash
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
{
SAFELIFE_HIP_LIB=$E/lib_advjc.so timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "advance_board or golden or patterns or side_effect_occupancy" 2>&1 | tail -2
for rep in 1 2; do
  timeout 300 python tools/occ_bench.py 2>&1 | grep advance
  SAFELIFE_HIP_LIB=$E/lib_advjc.so timeout 300 python tools/occ_bench.py 2>&1 | grep advance
  timeout 300 python tools/occ_bench.py append_spawn_25 8192 1000 2>&1 | grep advance
  SAFELIFE_HIP_LIB=$E/lib_advjc.so timeout 300 python tools/occ_bench.py append_spawn_25 8192 1000 2>&1 | grep advance
done
} > gpurun_out/r5ai_advjc.txt 2>&1
cat gpurun_out/r5ai_advjc.txt
