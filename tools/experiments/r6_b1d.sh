#!/bin/bash
# round 6: 6^3 layers with their weights through LDS -- bits, seam latency with / without, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_H16_WLDS=0

OUT=$R/gpurun_out/prof_r6b1d; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py > $OUT/log.txt 2>&1
grep "median call" $OUT/log.txt
cd $R; python tools/experiments/b1_timeline_report.py $OUT/trace | tee $OUT/timeline.txt | grep "h2_16\|span\|gmax"
python tools/experiments/dense_throughput.py
python tools/experiments/dense_throughput.py MI_GNINA_H16_WLDS=0
