#!/bin/bash
# Round 5, fifth GPU call: the tree (suite, soak, full bench line with extras) and timing-only builds of the C3 step
# kernel (what the phases are worth at the default launcher).
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
E=$PWD/tools/exp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r5e_pytest.txt
( timeout 200 python tools/soak.py 90 11 2>&1 | tail -4 ) > $O/r5e_soak.txt
for rep in 1 2 3; do
  for lib in sw skip1 skip2 skip3 skip4 skip7; do
    SAFELIFE_HIP_LIB=$E/lib_$lib.so timeout 300 python bench.py --steps 400 --warmup 40 --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib K=%d %.3f us/step frac %.3f' % (d['steps'], d['ms_per_step']*1e3, r['frac']))"
  done
done > $O/r5e_timing_only.txt 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5e_bench_k20_full.txt 2> $O/r5e_bench_k20_full.err
tail -c 600 $O/r5e_bench_k20_full.err
cat $O/r5e_pytest.txt $O/r5e_soak.txt $O/r5e_timing_only.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5e_bench_k20_full.txt').read().strip().splitlines()[-1])
print('K=20 %.3f us/step frac %.3f parity %s' % (d['ms_per_step']*1e3, d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_check')))
for k,v in sorted(d.get('extra',{}).items()):
    if isinstance(v,(int,float)): print('  %-60s %.4g' % (k,v))
    elif isinstance(v,str) and len(v)<100: print('  %-60s %s' % (k,v))
PY
