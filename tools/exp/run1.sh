#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for f in copy_bw launch_floor; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$f.hip -o /tmp/$f.bin && /tmp/$f.bin
done
python tools/exp/pipe_exp.py
for v in nt sc1 sc0sc1 ldnt; do
  SL_SLICES=1,2 SAFELIFE_HIP_LIB=tools/exp/lib_$v.so python tools/exp/pipe_exp.py
done
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --extras 0 --rollout 0
python bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0
} > gpurun_out/exp1.log 2>&1
tail -5 gpurun_out/exp1.log
