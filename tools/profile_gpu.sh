#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/*
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $BENCH > $OUT/trace_bench.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $OUT/pmc_inst -o p -- $BENCH > $OUT/pmc_inst.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -f csv -d $OUT/pmc_grbm -o p -- $BENCH > $OUT/pmc_grbm.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -f csv -d $OUT/pmc_occ -o p -- $BENCH > $OUT/pmc_occ.log 2>&1
find $OUT -name "*.csv" | head -40
tail -3 $OUT/pmc_sq.log | cut -c1-300
