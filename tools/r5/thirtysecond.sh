#!/bin/bash
# final tree: the episode-end pass of C5 -- kernel trace and counters; the headline at K = 20 and K = 400 (+ kernel trace)
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_se -- python $GRAFT_REPO_ROOT/tools/exp/se_pass.py > $GRAFT_REPO_ROOT/$O/prof_se.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(ls $O/prof_se/*/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db | head -14 > $O/r5s_se_pass_kernel_trace.txt
grep "pass:" $O/prof_se.log >> $O/r5s_se_pass_kernel_trace.txt
rm -rf $O/prof_se
bash tools/pmc_any.sh r5s_se_pass "occupancy|k_se_" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32" \
  "GRBM_GUI_ACTIVE" \
  -- python tools/exp/se_pass.py > /dev/null 2>&1
for k in 20 400 20 400; do
  w=$([ $k = 400 ] && echo 40 || echo 5)
  timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('K=%d %.3f us/step frac %.3f host %.2f us fences %s' % (d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3, d['config']['queue_fences'][:5]))"
done > $O/r5s_bench.txt 2>&1
timeout 900 python bench.py --steps 400 --warmup 40 > $O/r5s_bench_default.txt 2> /dev/null
cat $O/r5s_se_pass_kernel_trace.txt $O/r5s_bench.txt
