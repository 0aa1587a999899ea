#!/bin/bash
# Round 6, call g: staged regions (slhip_queues_stage / _go) and the first step without its system-scope acquire: queue tests, then the driver's bench command three times
cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "queue" 2>&1 | tail -4 ) > $O/r6g_pytest.txt
cat $O/r6g_pytest.txt
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 2>$O/r6g_bench_$i.err | tail -1 > $O/r6g_bench_$i.txt
  python - $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r6g_bench_%s.txt' % sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print('K=20: %.3f us/step frac %.3f | host %.2f | ' % (d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3) + ' '.join('%s %.2f' % (k[:-3], r[k]) for k in ('agent_fences_us','no_reset_us','k400_us','unstaged_us','k20_median_us','forced_gather_us','c5_with_side_effects_us') if r.get(k)), '| parity', d['cpu_baseline']['parity_check']['bit_exact'], d['config']['staged_steps'][:12])
PY
done
