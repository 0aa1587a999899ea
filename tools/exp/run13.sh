#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 | tail -1 | cut -c1-900
} > gpurun_out/exp13.log 2>&1
grep -v "amdgpu.ids" gpurun_out/exp13.log | tail -12
