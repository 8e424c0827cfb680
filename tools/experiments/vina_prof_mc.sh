#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_vina_mc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/experiments/vina_prof_driver_mc.py"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/pmc_a -o p -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/pmc_b -o p -- $CMD > $OUT/pmc_b.log 2>&1
grep -h "^[0-9]" $OUT/trace.log | tail -2
