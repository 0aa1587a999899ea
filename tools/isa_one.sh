#!/bin/bash
# ISA of ONE instantiation of the fused step kernel (seconds instead of the minutes the whole file takes):
#   tools/isa_one.sh "25, 25, true, false, false, true, true" [extra hipcc flags]   -> /tmp/isa_one.s + a summary
# template arguments: H, W, LDS_LUT, SPAWN, WRAP, LEAN, ONE.  The summary counts instructions by class between the
# kernel's s_barrier instructions (the phases of a single-step launch).
cd "$(dirname "$0")/.."
ARGS=${1:-"25, 25, true, false, false, true, true"}; shift
OUT=${ISA_OUT:-/tmp/isa_one.s}
cat > /tmp/isa_one.hip <<SRC
#define SL_ROWLANE_PART 99
#include "$PWD/safelife_amd/csrc/sl_rowlane.hip"
void *isa_one_kernel() { return (void *)sl::rl::k_env_rollout_rowlane<$ARGS>; }
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 \
    -Iinclude -Isafelife_amd/csrc -S --cuda-device-only "$@" -o $OUT /tmp/isa_one.hip || exit 1
python3 tools/isa_summary.py $OUT
