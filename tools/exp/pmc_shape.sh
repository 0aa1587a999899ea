#!/bin/bash
# PMC counters of the fused step on a synthetic pool of one board shape: tools/exp/pmc_shape.sh H W
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_shape_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; local h=$1 w=$2; shift 2
  rocprofv3 --pmc "$@" -d $OUT/$name -- python $ROOT/tools/exp/shape_bench.py $h $w > $OUT/$name.log 2>&1
  local db=$(ls $OUT/$name/*/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep -E "rollout" | grep -v "true, false>\|, false>(" | awk '{print $(NF-2), $(NF-1), $NF}' | sort | uniq
  rm -rf $OUT/$name; }
echo "== $1 x $2"
run a $1 $2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run b $1 $2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY
