#!/bin/bash
# round 6: gather with the pose's coordinates through LDS and its passes' loads a pass ahead; results and range flag written
# to pinned host memory by the reduction kernel -- full GPU suite, seam latency with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_OUT_COPY=1
OUT=$R/gpurun_out/prof_r6b1h; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py > $OUT/log.txt 2>&1
grep "median call" $OUT/log.txt
cd $R; python tools/experiments/b1_timeline_report.py $OUT/trace > $OUT/timeline.txt; head -8 $OUT/timeline.txt; tail -5 $OUT/timeline.txt
