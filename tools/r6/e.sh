#!/bin/bash
# Round 6, call e: what the level fetch of a reloading workgroup costs (timing-only build), and the driver's bench command on the new bench.py
cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do for v in 0 1; do
  SAFELIFE_HIP_LIB=$PWD/tools/exp/lib_reset$v.so SAFELIFE_HIP_LIB_ANY_ABI=1 timeout 300 python tools/exp/kfit.py 1 none 5 2>&1 | grep -v amdgpu.ids | tail -4
done; done > $O/r6e_timing_reset.txt 2>&1
cat $O/r6e_timing_reset.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/r6e_bench_k20.txt 2> $O/r6e_bench_k20.err
tail -5 $O/r6e_bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e_bench_k20.txt').read().strip().splitlines()[-1])
r=d['roofline']
print('value %.4g ms_per_step %.5f frac %.3f' % (d['value'], d['ms_per_step'], r['frac']))
print({k:r.get(k) for k in ('launches_per_step','agent_fences_us','no_reset_us','k400_us','k20_median_us','forced_gather_us','c5_with_side_effects_us','c5_with_side_effects_streams_us','life_occupancy_64x64_board_steps_per_s','traffic','traffic_agent_fences','host_enqueue_ms_per_step')})
print(d['config']['queue_ids'], d['config']['parallelism'][-140:], d['config']['episode_phase'])
print(d['cpu_baseline'])
print({k:v for k,v in d['extra'].items() if 'error' in k})
PY
