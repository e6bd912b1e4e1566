#!/bin/bash
# Counter evidence for the update kernels (K12 chain / two-XCD persistent kernel, K14, K15): separate rocprofv3 --pmc passes
# per MI355X_MICROARCH.md (TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ: 8 slots), the program directly after `--`.
#   bash tools/update_pmc.sh r03 C2 C3 C4 C5   ->  gpurun_out/prof/<tag>_update_pmc.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof/${TAG}_update_pmc.csv
: > $OUT
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
for C in "$@"; do
  i=0
  for P in "${PASSES[@]}"; do
    rm -rf /tmp/upmc_$i
    rocprofv3 --pmc $P --output-format csv -d /tmp/upmc_$i -- python3 $R/tools/update_pmc_driver.py --config $C > /tmp/upmc_$i.log 2>&1
    echo "$C pass $i ($P) rc=$?"
    python3 $R/tools/summarize_update_pmc.py /tmp/upmc_$i $C >> $OUT
    i=$((i+1))
  done
done
cat $OUT
