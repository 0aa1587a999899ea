#!/bin/bash
# Build A/B variants of the library: tools/exp/build_variants.sh name "-DFLAGS" [name "-DFLAGS" ...]
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 -Iinclude -DSL_EXPERIMENTS"
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FL $defs safelife_amd/csrc/*.hip -o tools/exp/lib_$name.so && echo built $name ) &
done
wait
