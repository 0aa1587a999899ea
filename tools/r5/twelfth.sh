#!/bin/bash
# Round 5, final tree: full suite, soaks, the driver's bench command, K = 400, kernel trace + counters, C5 pass alone
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/r5m_pytest.txt
( timeout 200 python tools/soak.py 90 31 2>&1 | tail -3 ) > $O/r5m_soak.txt
( timeout 200 python tools/soak_prims.py 40 5 2>&1 | tail -2 ) >> $O/r5m_soak.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r5m_bench_k20.txt 2> $O/r5m_bench_k20.err
timeout 1200 python bench.py --steps 400 --warmup 40 --extras 0 > $O/r5m_bench_default.txt 2> $O/r5m_bench_default.err
( timeout 300 python tools/exp/se_pass.py 2>&1 | tail -1 ) > $O/r5m_se_pass.txt
bash tools/prof_run.sh r5m_queues4_none --stream-leg 0 > /dev/null 2>&1
cat $O/r5m_pytest.txt $O/r5m_soak.txt $O/r5m_se_pass.txt
python - <<'PY'
import json
for f in ('r5m_bench_k20','r5m_bench_default'):
    d=json.loads(open('gpurun_out/%s.txt'%f).read().strip().splitlines()[-1])
    print(f, 'K=%d %.3f us/step frac %.3f parity %s' % (d['steps'], d['ms_per_step']*1e3, d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_check')))
    for k,v in sorted(d.get('extra',{}).items()):
        if isinstance(v,(int,float)) and ('us_per' in k or 'board_steps' in k): print('  %-60s %.4g' % (k,v))
PY
head -6 $O/r5m_queues4_none_kernel_trace.txt; grep -E "INSTS_VALU|INSTS_LDS|INSTS_SALU|SQ_WAVES" $O/r5m_queues4_none_pmc.txt | tail -4
