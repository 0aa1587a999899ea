#!/bin/bash
# PMC counters of tools/ab_step.py in one mode: tools/exp/pmc_mode.sh <mode>   (SL_MODE: plain | obs15 | obs19 | wrap)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_mode_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; local mode=$1; shift
  SL_MODE=$mode rocprofv3 --pmc "$@" -d $OUT/$name -- python $ROOT/tools/ab_step.py > $OUT/$name.log 2>&1
  local db=$(ls $OUT/$name/*/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep -E "rollout" | awk '{print $(NF-2), $(NF-1), $NF}' | sort | uniq | awk '$2 > 100'
  grep "us/step" $OUT/$name.log | tail -1
  rm -rf $OUT/$name; }
echo "== $1"
run a $1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR
run b $1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
