#!/bin/bash
# Round 6, call b: chained queue stepping (SL_QUEUES_CHAINED) -- parity of the queue tests, then the region-length fit with and without it
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -k "queue_stepping_vs_oracle or queue_steps_write or release_free" 2>&1 | tail -8 ) > $O/r6b_pytest.txt
cat $O/r6b_pytest.txt
for ch in 0 1 0 1; do timeout 300 python tools/exp/kfit.py 0 none 5 $ch 2>&1 | grep -v amdgpu.ids | tail -9; done > $O/r6b_kfit.txt 2>&1
for ch in 0 1; do timeout 300 python tools/exp/kfit.py 1 none 5 $ch 2>&1 | grep -v amdgpu.ids | tail -9; done >> $O/r6b_kfit.txt 2>&1
for nq in 1 2; do timeout 300 python tools/exp/kfit.py 0 none 5 1 $nq 2>&1 | grep -v amdgpu.ids | tail -9; done >> $O/r6b_kfit.txt 2>&1
cat $O/r6b_kfit.txt
