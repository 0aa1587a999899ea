#!/bin/bash
# Collect PMC counters for the bench kernel in separate passes (rocprofv3 --pmc only; no tracing),
# as the gpurun rules and MI355X_MICROARCH.md (PMC slots: SQ 8, TCC 4, GRBM 2) require.
# usage: tools/pmc_run.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --cpu-baseline 0 --extras 0 --rollout 0 "${BARGS[@]}" > $OUT/$name.log 2>&1
  local db=$(ls $OUT/$name/*/*_results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/prof_summary.py $db --pmc | grep -E "counter|rollout|advance" > $OUT/$name.txt
  cat $OUT/$name.txt
  rm -rf $OUT/$name          # raw databases are large; only the summaries travel back
}
BARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY
run tcc_rd FETCH_SIZE
run tcc_wr WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
