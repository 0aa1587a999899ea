#!/bin/bash
# Round 6, call a: the tree as round 5 left it -- suite, region length fit (fixed cost of the timed region), agent-mode PMC traffic
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > $O/r6a_pytest.txt
for s in 0 1; do timeout 300 python tools/exp/kfit.py $s none 7; done > $O/r6a_kfit.txt 2>&1
timeout 300 python tools/exp/kfit.py 0 agent 7 >> $O/r6a_kfit.txt 2>&1
bash tools/pmc_run.sh r6a --queues 4 --queue-fences agent --stream-leg 0 > $O/r6a_pmc_agent.txt 2>&1
cat $O/r6a_pytest.txt $O/r6a_kfit.txt
