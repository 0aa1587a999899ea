#!/bin/bash
# PMC counters of ANY command, one rocprofv3 --pmc pass per group of counters (no tracing in those passes, as the
# gpurun rules and MI355X_MICROARCH.md require); per-kernel averages of every pass are appended to
# gpurun_out/<tag>_pmc.txt.
# usage: tools/pmc_any.sh <tag> <kernel-name regex> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <command...>
set -u
TAG=$1; KRE=$2; shift 2
PASSES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
RES=$ROOT/gpurun_out/${TAG}_pmc.txt
: > $RES
cd /tmp && export TMPDIR=/tmp
i=0
for counters in "${PASSES[@]}"; do
  i=$((i + 1))
  echo "## pass $i: $counters" >> $RES
  ( cd $ROOT && rocprofv3 --pmc $counters -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1 )
  db=$(ls $OUT/p$i/*/*_results.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    python $ROOT/tools/prof_summary.py $db --pmc | grep -E "^kernel +counter|$KRE" | grep -v "^#" >> $RES
  else
    echo "(no database: $(tail -2 $OUT/p$i.log | tr '\n' ' '))" >> $RES
  fi
  grep -E "^\{|us/step|pass:" $OUT/p$i.log | tail -1 | cut -c1-400 >> $RES
  rm -rf $OUT/p$i
done
cat $RES
