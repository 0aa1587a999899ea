#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "test_queue_stepping_vs_oracle and append_spawn" 2>&1 | grep -v "^$" | tail -30 ) > gpurun_out/r5g_dbg.txt
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "planted or pool_refresh or recovers" 2>&1 | tail -5 ) >> gpurun_out/r5g_dbg.txt
cat gpurun_out/r5g_dbg.txt
