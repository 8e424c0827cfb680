#!/bin/bash
# round 6: producer / consumer form of the 6^3 layers' ring kernel -- bits, seam latency with / without, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_concurrency.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_H16_WLDS=9
timeout 300 python tools/experiments/seam_b1_ensemble.py
OUT=$R/gpurun_out/prof_r6pc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py > $OUT/log.txt 2>&1
cd $R; python tools/experiments/b1_timeline_report.py $OUT/trace | grep "pc_kernel\|ring" | awk '{print $2, $NF}' | tr '\n' ' '
