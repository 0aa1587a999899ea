#!/bin/bash
# Round 5, fourth GPU call: prologue variants of the step kernel (scalar wave index, issue priority until the loads are
# out) and the episode-end pass without LDS counters / with the lanes' cached jumps -- parity first, then timings.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
E=$PWD/tools/exp
for lib in occ occjc; do
  echo "== lib_$lib"
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -4
done > $O/r5d_occ_pytest.txt 2>&1
for rep in 1 2; do
  echo -n "in-tree: "
  timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  for lib in occjc occ; do
    echo -n "lib_$lib: "
    SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  done
done > $O/r5d_se_pass.txt 2>&1
for rep in 1 2 3; do
  for lib in base2 sw swp1 swp3; do
    for k in 400 20; do
      w=$([ $k = 400 ] && echo 40 || echo 5)
      SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib K=%d %.3f us/step frac %.3f host %.2f us' % (d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3))"
    done
  done
done > $O/r5d_ab_prologue.txt 2>&1
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "goal_word_cache" 2>&1 | tail -5 ) > $O/r5d_pytest.txt
( timeout 200 python tools/soak.py 100 7 2>&1 | tail -4 ) > $O/r5d_soak.txt
cat $O/r5d_occ_pytest.txt $O/r5d_se_pass.txt $O/r5d_ab_prologue.txt $O/r5d_pytest.txt $O/r5d_soak.txt
