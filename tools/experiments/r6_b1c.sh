#!/bin/bash
# round 6: what a d16 launch of a per-pose call consists of -- timelines with the kernel's timing switches
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
for V in 2 6 14 78; do
OUT=$R/gpurun_out/prof_r6b1c_$V; rm -rf $OUT; mkdir -p $OUT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py MI_GNINA_D16_DBG=$V > $OUT/log.txt 2>&1)
echo "== MI_GNINA_D16_DBG=$V: $(grep 'median call' $OUT/log.txt)"
python tools/experiments/b1_timeline_report.py $OUT/trace | grep "d16\|span"
done
