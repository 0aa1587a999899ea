#!/bin/bash
# Register / scratch / code-size summary of the fused step kernels (compiles sl_rowlane.hip to ISA in /tmp):
#   tools/kstat.sh [extra hipcc flags]      -> one line per k_env_rollout_rowlane instantiation of 25x25 and 64x64
cd "$(dirname "$0")/.."
OUT=${KSTAT_OUT:-/tmp/kstat_rowlane.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 \
    -Iinclude -S --cuda-device-only "$@" -o $OUT safelife_amd/csrc/sl_rowlane.hip 2>/dev/null || exit 1
python3 - $OUT <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_ZN2sl2rl\w+):.*?; codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", txt, re.S | re.M):
    name = m.group(1)
    if "rollout_rowlane" in name or "advance" in name or "occupancy" in name:
        short = re.sub(r"EEv.*", "", name.replace("_ZN2sl2rl", ""))
        print("%-60s code %6s B  vgpr %3s  scratch %4s  occupancy %s" % (short, m.group(2), m.group(3), m.group(4), m.group(5)))
PY
