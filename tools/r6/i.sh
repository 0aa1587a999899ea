#!/bin/bash
# Round 6, call i: ADVICE r5 fixes (slot reuse tied to flushes, recovery log, goal-cache size) -- refresh / recovery / queue / side-effect tests, then the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "pool or recover or release_free or side_effect or goal" 2>&1 | tail -8 ) > $O/r6i_pytest_sel.txt
cat $O/r6i_pytest_sel.txt
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/r6i_pytest.txt
cat $O/r6i_pytest.txt
