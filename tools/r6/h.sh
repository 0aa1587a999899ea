#!/bin/bash
# Round 6, call h: resets that copy the level as an episode starts on it (sl_env_batch.pool_ready): the whole GPU suite, then A/B on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r6h_pytest.txt
cat $O/r6h_pytest.txt
for rep in 1 2; do for pr in 0 1; do
  echo "SAFELIFE_POOL_READY=$pr"
  SAFELIFE_POOL_READY=$pr KFIT_NO_HEAD=1 timeout 300 python tools/exp/kfit.py 1 none 5 2>&1 | grep -v amdgpu.ids | grep -E "K= 20|K=400|elapsed"
done; done > $O/r6h_ab_pool_ready.txt 2>&1
cat $O/r6h_ab_pool_ready.txt
