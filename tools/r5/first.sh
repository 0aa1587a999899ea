#!/bin/bash
# Round 5, first GPU call: the tree as round 4 left it -- suite, bench lines, issue-side counters of the C3 step kernel,
# and the episode-end pass of C5 (kernel trace + counters).
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > $O/r5a_pytest.txt
for k in 20 400 20 400; do
  w=$([ $k = 400 ] && echo 40 || echo 5)
  timeout 300 python bench.py --steps $k --warmup $w --extras 0 --rollout 0 --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('K=%d %.3f us/step frac %.3f host %.2f us fences %s' % (d['steps'], d['ms_per_step']*1e3, r['frac'], r['host_enqueue_ms_per_step']*1e3, d['config']['queue_fences'][:5]))"
done > $O/r5a_bench.txt 2>&1
BENCH="python bench.py --steps 100 --warmup 10 --cpu-baseline 0 --extras 0 --rollout 0 --stream-leg 0"
bash tools/pmc_any.sh r5a_c3_issue "rollout" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
  "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
  "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES" \
  "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2 SQ_INSTS_VSKIPPED" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  -- $BENCH > /dev/null 2>&1
# the episode-end pass of C5 (tools/exp/se_pass.py): kernel trace, then counters
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_se -- python $GRAFT_REPO_ROOT/tools/exp/se_pass.py > $GRAFT_REPO_ROOT/$O/prof_se.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(ls $O/prof_se/*/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db | head -14 > $O/r5a_se_pass_kernel_trace.txt
grep "pass:" $O/prof_se.log >> $O/r5a_se_pass_kernel_trace.txt
rm -rf $O/prof_se
bash tools/pmc_any.sh r5a_se_pass "occupancy|k_se_" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_INT64 SQ_LEVEL_WAVES" \
  "GRBM_GUI_ACTIVE" \
  -- python tools/exp/se_pass.py > /dev/null 2>&1
rocprofv3 -L > $O/r5a_counters_list.txt 2>&1
ls -la $O | tail -20
