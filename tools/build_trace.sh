#!/bin/bash
# Build a -DSL_TRACE copy of the library for tools/trace_phases.py (SAFELIFE_HIP_LIB=tools/lib_trace.so)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 \
    -DSL_TRACE -Iinclude safelife_amd/csrc/*.hip -o tools/lib_trace.so
