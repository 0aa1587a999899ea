#!/bin/bash
# the tree: whole GPU suite, both soaks, the bench at K = 20
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/soak.py 2>&1 | tail -3
timeout 900 python tools/soak_prims.py 2>&1 | tail -3
} > $O/r5af_suite.txt 2>&1
cat $O/r5af_suite.txt
bash tools/r5/twentyfirst.sh
