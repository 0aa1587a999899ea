#!/bin/bash
# TIMING-ONLY build (WRONG RESULTS BY CONSTRUCTION): a workgroup that reloads a level fetches nothing of it from the pool
# (-DSL_TIMING_RESET=1) -- the most a prefetch of the next level under the step could save.  -> tools/exp/lib_reset0.so / lib_reset1.so
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 -Iinclude -DSL_DEV_SHAPES"
W=/tmp/safelife_timing_reset; mkdir -p $W
for f in sl_abi sl_generic sl_side_effects sl_aql; do
  /opt/rocm/bin/hipcc $FL -c safelife_amd/csrc/$f.hip -o $W/$f.o &
done
wait
for v in 0 1; do
  ( /opt/rocm/bin/hipcc $FL -DSL_TIMING_RESET=$v -c safelife_amd/csrc/sl_rowlane.hip -o $W/rl_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $W/rl_$v.o $W/sl_abi.o $W/sl_generic.o $W/sl_side_effects.o $W/sl_aql.o -o tools/exp/lib_reset$v.so && echo built lib_reset$v.so ) &
done
wait
