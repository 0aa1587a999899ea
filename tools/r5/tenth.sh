#!/bin/bash
# DPP butterflies in the draws' outcome packing: parity, then the pass alone and C4's share, against the in-tree build
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
O=gpurun_out
mkdir -p $O
( SAFELIFE_HIP_LIB=$E/lib_dpp.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect or advance_board or append_spawn or goal_word or full_size" 2>&1 | tail -3 ) > $O/r5k_pytest.txt 2>&1
for rep in 1 2 3; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "lib_dpp: "; SAFELIFE_HIP_LIB=$E/lib_dpp.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  for lib in "" $E/lib_dpp.so; do
    SAFELIFE_HIP_LIB=$lib timeout 300 python bench.py --pool append_spawn_25 --steps 400 --warmup 40 --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('C4 ${lib:-in-tree} K=%d %.3f us/step' % (d['steps'], d['ms_per_step']*1e3))"
  done
done > $O/r5k_dpp.txt 2>&1
cat $O/r5k_pytest.txt $O/r5k_dpp.txt
