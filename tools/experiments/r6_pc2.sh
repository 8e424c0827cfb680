#!/bin/bash
# round 6: the fixed part of a conv3d_h2_16_pc_kernel launch (MI_GNINA_H2_DBG 4096 = no chunks, 8192 = return after the first prologue lines)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
for V in 0 4096 8192; do
OUT=$R/gpurun_out/prof_r6pc2_$V; rm -rf $OUT; mkdir -p $OUT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py MI_GNINA_H2_DBG=$V > $OUT/log.txt 2>&1)
echo "== MI_GNINA_H2_DBG=$V: $(grep 'median call' $OUT/log.txt) pc kernels: $(python tools/experiments/b1_timeline_report.py $OUT/trace | grep 'pc_kernel' | awk '{printf "%s ", $NF}') gmax/heads: $(python tools/experiments/b1_timeline_report.py $OUT/trace | grep 'gmax\|fc_heads' | awk '{printf "%s ", $NF}')"
done
