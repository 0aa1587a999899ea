#!/bin/bash
# A/B builds of the library that differ in the row kernels only: tools/exp/build_variants.sh name "-DFLAGS" [name "-DFLAGS" ...] -> tools/exp/lib_<name>.so (compare with tools/exp/abn.sh)
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 -Iinclude"
mkdir -p /tmp/safelife_variants
for f in sl_abi sl_generic sl_side_effects sl_rowlane_b sl_rowlane_c; do
  [ -f /tmp/safelife_variants/$f.o ] || /opt/rocm/bin/hipcc $FL -c safelife_amd/csrc/$f.hip -o /tmp/safelife_variants/$f.o &
done
wait
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FL $defs -c safelife_amd/csrc/sl_rowlane.hip -o /tmp/safelife_variants/rl_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc /tmp/safelife_variants/rl_$name.o /tmp/safelife_variants/sl_abi.o /tmp/safelife_variants/sl_generic.o /tmp/safelife_variants/sl_side_effects.o /tmp/safelife_variants/sl_rowlane_b.o /tmp/safelife_variants/sl_rowlane_c.o -o tools/exp/lib_$name.so && echo built $name ) &
done
wait
