#!/bin/bash
# episode-end pass: tree (2.25 waves per SIMD) against lib_occ2 (4): where does the extra occupancy go?
cd $GRAFT_REPO_ROOT
E=$PWD/tools/exp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
P2="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CU_CYCLES"
P3="SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQC_ICACHE_BUSY_CYCLES"
P4="GRBM_GUI_ACTIVE"
bash tools/pmc_any.sh r5q_tree "k_occupancy_rowlane<64, 64, [24]>" "$P1" "$P2" "$P3" "$P4" -- python tools/exp/se_pass.py > /dev/null 2>&1
SAFELIFE_HIP_LIB=$E/lib_occ2.so bash tools/pmc_any.sh r5q_occ2 "k_occupancy_rowlane<64, 64, [24]>" "$P1" "$P2" "$P3" "$P4" -- python tools/exp/se_pass.py > /dev/null 2>&1
for t in tree occ2; do echo "==== $t"; cut -c40-200 gpurun_out/r5q_${t}_pmc.txt; done
