#!/bin/bash
# round 5, GPU call 45: the voxelizer's instruction mix after the SALU trims (PMC passes of the short bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/prof_r5vox; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES --kernel-trace -f csv -d $OUT/pmc_a -o p -- $BENCH > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $OUT/pmc_b -o p -- $BENCH > $OUT/pmc_b.log 2>&1
tail -2 $OUT/pmc_a.log | cut -c1-300; tail -2 $OUT/pmc_b.log | cut -c1-300
cd $R; python tools/pmc_summary.py gpurun_out/prof_r5vox 2>&1 | awk '/^voxelize_tiles/{f=1} /^[a-z_]/{if(!/^voxelize_tiles/)f=0} f' | head -40
