#!/bin/bash
# the plain single-step kernels without a goal image in LDS (dev builds: 25x25 and 64x64 shapes only): parity, then
# C3 / C4 / C5 shares against the in-tree build
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
O=gpurun_out
mkdir -p $O
( SAFELIFE_HIP_LIB=$E/lib_ng.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "(goal_word and not append_still) or full_size or queue_stepping_vs or pool_refresh or recovers or env_batch_vs or test_env_trace" 2>&1 | tail -4 ) > $O/r5l_pytest.txt 2>&1
( SAFELIFE_HIP_LIB=$E/lib_ng.so timeout 200 python tools/soak.py 90 21 2>&1 | tail -3 ) > $O/r5l_soak.txt 2>&1
for rep in 1 2; do
  for lib in "" $E/lib_ng.so $E/lib_ng2.so; do
    for cfg in "prune_still_25 8192" "append_spawn_25 8192" "navigation_64 4096"; do
      set -- $cfg
      SAFELIFE_HIP_LIB=$lib timeout 300 python bench.py --pool $1 --envs $2 --steps 400 --warmup 40 --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 ${lib:-in-tree} K=%d %.3f us/step' % (d['steps'], d['ms_per_step']*1e3))"
    done
  done
done > $O/r5l_ab.txt 2>&1
cat $O/r5l_pytest.txt $O/r5l_soak.txt $O/r5l_ab.txt
