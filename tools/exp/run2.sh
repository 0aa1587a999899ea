#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python tools/exp/sync_exp.py
SL_SPIN=1 python tools/exp/sync_exp.py
HSA_ENABLE_INTERRUPT=0 python tools/exp/sync_exp.py
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/icache.hip -o /tmp/icache.bin 2>/dev/null && /tmp/icache.bin
} > gpurun_out/exp2.log 2>&1
grep -v "amdgpu.ids\|warning\|\^" gpurun_out/exp2.log | tail -60
