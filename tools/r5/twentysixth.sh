#!/bin/bash
# paired deal loop (tree): parity of everything that draws, the pass, C5 / C4 stepping
cd $GRAFT_REPO_ROOT
{
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect or advance_board or navigation or golden or trace or spawn" 2>&1 | tail -3
for rep in 1 2; do timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1; done
timeout 300 python tools/exp/c5_steps.py navigation_64 4096 4 2>&1 | grep "us/step" | tail -2
timeout 300 python tools/exp/c5_steps.py append_spawn_25 8192 4 2>&1 | grep "us/step" | tail -2
timeout 300 python tools/occ_bench.py 2>&1 | grep -v amdgpu
} > gpurun_out/r5aa_pairs.txt 2>&1
cat gpurun_out/r5aa_pairs.txt
