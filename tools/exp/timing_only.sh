#!/bin/bash
# TIMING-ONLY builds of the fused step kernel (WRONG RESULTS BY CONSTRUCTION): what the phases of the C3 step are worth at
# the bench's default launcher.  -DSL_TIMING_SKIP=mask compiles out: 1 the CA pass, 2 the row scores, 4 the leaders' work
# (barriers, loads, stores, record write-back and the launch boundary stay).  -DSL_DEV_SHAPES keeps the 25x25 and 64x64
# shapes only (a quarter of the compile time).
#   tools/exp/timing_only.sh            -> tools/exp/lib_cur.so, lib_skip1.so, lib_skip2.so, lib_skip3.so, lib_skip4.so, lib_skip7.so
# then, on the GPU box:
#   for lib in cur skip1 skip2 skip3 skip4 skip7; do
#     SAFELIFE_HIP_LIB=$PWD/tools/exp/lib_$lib.so python bench.py --steps 400 --warmup 40 --extras 0 --rollout 0 --cpu-baseline 0
#   done
# Round 5 (profiles/round5_f_timing_only.txt, one box, us per C3 step at K = 400): full 6.36-6.47 | no CA 5.69-5.75 |
# no scores 6.19-6.27 | neither 5.23-5.34 | no leader work 6.05-6.15 | none of the three 4.73-4.83.
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 -Iinclude -DSL_DEV_SHAPES"
W=/tmp/safelife_timing_only; mkdir -p $W
for f in sl_abi sl_generic sl_side_effects sl_aql; do
  /opt/rocm/bin/hipcc $FL -c safelife_amd/csrc/$f.hip -o $W/$f.o &
done
wait
for v in 0 1 2 3 4 7; do
  name=$([ $v = 0 ] && echo cur || echo skip$v)
  ( /opt/rocm/bin/hipcc $FL -DSL_TIMING_SKIP=$v -c safelife_amd/csrc/sl_rowlane.hip -o $W/rl_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $W/rl_$name.o $W/sl_abi.o $W/sl_generic.o $W/sl_side_effects.o $W/sl_aql.o -o tools/exp/lib_$name.so && echo built lib_$name.so ) &
done
wait
