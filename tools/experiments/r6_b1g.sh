#!/bin/bash
# round 6: what a chunk of conv3d_h2_16_ring_kernel consists of -- timelines with its timing switches (MI_GNINA_H2_DBG bits 256..2048)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
for V in 0 256 1024 512 1536 2048 3840; do
OUT=$R/gpurun_out/prof_r6b1g_$V; rm -rf $OUT; mkdir -p $OUT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py MI_GNINA_H2_DBG=$V > $OUT/log.txt 2>&1)
echo "== MI_GNINA_H2_DBG=$V: $(grep 'median call' $OUT/log.txt) ring kernels: $(python tools/experiments/b1_timeline_report.py $OUT/trace | grep 'ring' | awk '{printf "%s ", $NF}')"
done
