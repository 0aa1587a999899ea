#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
echo "== OLD"; (cd tools/exp/old_tree && SL_SLICES=1,2 python tools/exp/pipe_exp.py | grep "threaded=0")
echo "== NEW"; SL_SLICES=1,2 python tools/exp/pipe_exp.py | grep "threaded=0"
done
} > gpurun_out/exp10.log 2>&1
grep -v "amdgpu.ids" gpurun_out/exp10.log | tail -40
