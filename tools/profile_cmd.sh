#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + two PMC passes of an arbitrary command.
# Usage: tools/profile_cmd.sh <tag> <command...>   -> gpurun_out/prof_<tag>/{trace,pmc_mfma,pmc_mem}/*.csv
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/pmc_mfma -o p -- "$@" > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -f csv -d $OUT/pmc_inst -o p -- "$@" > $OUT/pmc_inst.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o p -- "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o p -- "$@" > $OUT/pmc_write.log 2>&1
find $OUT -name "*stats.csv" | head
