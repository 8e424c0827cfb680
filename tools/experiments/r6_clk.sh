#!/bin/bash
# round 6: which clocks does the chip run a loop of per-pose calls at?  (rocm-smi polled next to the loop -- reads only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -6
echo "== per-pose loop"
(timeout 300 python tools/experiments/seam_b1_ensemble.py > /tmp/seam.txt 2>&1) &
PID=$!
sleep 8
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -2 | tr '\n' ' '; rocm-smi --showuse 2>&1 | grep -i "busy" | head -1; sleep 1; done
wait $PID; cat /tmp/seam.txt
echo "== 1,024-pose steps"
(timeout 300 python tools/experiments/dense_throughput.py > /tmp/dense.txt 2>&1) &
PID=$!
sleep 7
for i in 1 2 3; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -2 | tr '\n' ' '; echo; sleep 0.5; done
wait $PID; tail -1 /tmp/dense.txt
