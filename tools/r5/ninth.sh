#!/bin/bash
# episode-end pass: counters flushed into the lane's own row of the output by plain read-modify-write (no LDS counters,
# no atomics) -- parity, then the pass alone
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
O=gpurun_out
mkdir -p $O
for lib in rmw rmwnojc; do
  echo "== lib_$lib"
  SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -3
done > $O/r5j_occ_pytest.txt 2>&1
for rep in 1 2; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  for lib in nojc rmw rmwnojc; do
    echo -n "lib_$lib: "
    SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  done
done > $O/r5j_se_pass.txt 2>&1
cat $O/r5j_occ_pytest.txt $O/r5j_se_pass.txt
