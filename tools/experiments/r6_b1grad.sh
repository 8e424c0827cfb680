#!/bin/bash
# round 6: timeline of a B = 1 gradient call (default2017, then gnina's default ensemble)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
for W in default2017 ensemble; do
python tools/experiments/b1_grad_timeline.py $W
OUT=$R/gpurun_out/prof_r6b1grad_$W; rm -rf $OUT; mkdir -p $OUT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_grad_timeline.py $W > $OUT/log.txt 2>&1)
echo "== $W under the tracer: $(grep 'median call' $OUT/log.txt)"
python tools/experiments/b1_grad_report.py $OUT/trace > $OUT/timeline.txt; cat $OUT/timeline.txt | head -150
done
