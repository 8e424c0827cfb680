#!/bin/bash
# gpurun: kernel trace + two PMC passes of the Vina BFGS kernel -> gpurun_out/prof_vina/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_vina
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/experiments/vina_prof_driver.py"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/pmc_a -o p -- $CMD > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH --kernel-trace -f csv -d $OUT/pmc_b -o p -- $CMD > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_IFETCH_LEVEL SQ_WAVES --kernel-trace -f csv -d $OUT/pmc_c -o p -- $CMD > $OUT/pmc_c.log 2>&1
ls -R $OUT | head -30
tail -3 $OUT/pmc_b.log $OUT/pmc_c.log
