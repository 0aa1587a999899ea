#!/bin/bash
# Round 5, sixth GPU call: the tree (suite; the queue test that failed last time says why now), timing-only builds,
# C5 with the episode-end pass beside the steps, the K = 20 timeline.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
E=$PWD/tools/exp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/r5f_pytest.txt
for rep in 1 2 3; do
  for lib in cur skip1 skip2 skip3 skip4 skip7; do
    SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python bench.py --steps 400 --warmup 40 --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib K=%d %.3f us/step frac %.3f' % (d['steps'], d['ms_per_step']*1e3, r['frac']))"
  done
done > $O/r5f_timing_only.txt 2>&1
( timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1 ) > $O/r5f_se_pass.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5f_bench_k20_full.txt 2> $O/r5f_bench_k20_full.err
( SL_AQL_TIMELINE=1 SL_BENCH_DEBUG=1 timeout 300 python bench.py --steps 20 --warmup 5 --extras 0 --rollout 0 --cpu-baseline 0 2>&1 | tail -60 ) > $O/r5f_k20_timeline.txt
cat $O/r5f_pytest.txt $O/r5f_timing_only.txt $O/r5f_se_pass.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5f_bench_k20_full.txt').read().strip().splitlines()[-1])
print('K=20 %.3f us/step frac %.3f parity %s' % (d['ms_per_step']*1e3, d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_check')))
for k,v in sorted(d.get('extra',{}).items()):
    if isinstance(v,(int,float)): print('  %-60s %.4g' % (k,v))
PY
