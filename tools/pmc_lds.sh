#!/bin/bash
# Usage: tools/pmc_lds.sh <outdir-under-gpurun_out> [prof_run args...]  -- LDS / issue counters of one case
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
PASSES=(
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -s KILL 90 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
