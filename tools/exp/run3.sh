#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for sl in 1 2; do
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --extras 0 --rollout 0 --slices $sl | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('slices', r['launches_per_step'], 'K=20 ms_per_step %.2f us  launch %.2f us frac %.3f frac_wall %.3f' % (d['ms_per_step']*1e3, r['launch_ms']*1e3, r['frac'], r['frac_wall']))"
done
python bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0 --slices $sl | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('slices', r['launches_per_step'], 'K=400 ms_per_step %.2f us  launch %.2f us frac %.3f frac_wall %.3f' % (d['ms_per_step']*1e3, r['launch_ms']*1e3, r['frac'], r['frac_wall']))"
done
python bench.py --steps 20 --warmup 5 --slices 2 | tail -1
} > gpurun_out/exp3.log 2>&1
grep -v "amdgpu.ids" gpurun_out/exp3.log | tail -20
