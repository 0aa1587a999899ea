#!/bin/bash
# two-slot occupancy counters (dev lib occ2) against the tree: parity, the pass alone, life_occupancy
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
O=gpurun_out
mkdir -p $O
{
python tools/r5/pool_probe.py 2>&1 | grep "us/step"
echo "== parity, lib_occ2"
SAFELIFE_HIP_LIB=$E/lib_occ2.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "occupancy or side_effect" 2>&1 | tail -3
echo "== parity, tree"
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "colour_classes" 2>&1 | tail -3
for rep in 1 2; do
  echo -n "in-tree: "; timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "lib_occ2: "; SAFELIFE_HIP_LIB=$E/lib_occ2.so timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1
  echo -n "in-tree: "; timeout 300 python tools/occ_bench.py 2>&1 | head -1
  echo -n "lib_occ2: "; SAFELIFE_HIP_LIB=$E/lib_occ2.so timeout 300 python tools/occ_bench.py 2>&1 | head -1
done
} > $O/r5p_occ2.txt 2>&1
cat $O/r5p_occ2.txt
