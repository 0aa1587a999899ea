#!/bin/bash
# Round 6, call f: what the fixed cost of a timed region is made of -- the first step's system-scope acquire, the stream-side synchronize
cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do
  timeout 300 python tools/exp/kfit.py 0 none 5 2>&1 | grep -v amdgpu.ids | grep -E "K=  1|K=  5|K= 20|elapsed"
  KFIT_NO_HEAD=1 timeout 300 python tools/exp/kfit.py 0 none 5 2>&1 | grep -v amdgpu.ids | grep -E "K=  1|K=  5|K= 20|elapsed"
  KFIT_NO_HEAD=1 KFIT_NO_TORCH_SYNC=1 timeout 300 python tools/exp/kfit.py 0 none 5 2>&1 | grep -v amdgpu.ids | grep -E "K=  1|K=  5|K= 20|elapsed"
  KFIT_NO_HEAD=1 timeout 300 python tools/exp/kfit.py 1 none 5 2>&1 | grep -v amdgpu.ids | grep -E "K=  1|K=  5|K= 20|elapsed"
done > $O/r6f_fixed_cost.txt 2>&1
cat $O/r6f_fixed_cost.txt
