#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python -m pytest tests -m gpu -x -q -k "sliced" 2>&1 | tail -1; done
SL_SLICES=1,2 python tools/exp/pipe_exp.py | grep "threaded=0"
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --extras 0 --rollout 0 --slices 2 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('slices', r['launches_per_step'], 'K=20 ms_per_step %.2f us  launch %.2f us host %.2f frac %.3f frac_wall %.3f' % (d['ms_per_step']*1e3, r['launch_ms']*1e3, r['host_enqueue_ms_per_step']*1e3, r['frac'], r['frac_wall']))"
done
python bench.py --steps 400 --warmup 40 --cpu-baseline 0 --extras 0 --rollout 0 --slices 2 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('slices', r['launches_per_step'], 'K=400 ms_per_step %.2f us  launch %.2f us host %.2f frac %.3f frac_wall %.3f' % (d['ms_per_step']*1e3, r['launch_ms']*1e3, r['host_enqueue_ms_per_step']*1e3, r['frac'], r['frac_wall']))"
} > gpurun_out/exp11.log 2>&1
grep -v "amdgpu.ids" gpurun_out/exp11.log | tail -40
