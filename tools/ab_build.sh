#!/bin/bash
# Build a variant of the library for same-box A/B runs: tools/ab_build.sh <tag> [-DSL_...=..]  ->  tools/lib_<tag>.so
# (run with SAFELIFE_HIP_LIB=tools/lib_<tag>.so; one hipcc per source, side by side)
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
TMP=$(mktemp -d /tmp/sl_ab_XXXX)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8"
for src in safelife_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o $TMP/$(basename $src).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $TMP/*.o -o tools/lib_$TAG.so
rm -rf $TMP
ls -la tools/lib_$TAG.so
