#!/bin/bash
# round 6: first run of the stationary-weights first conv (bits and time) + the concurrency tests + B = 1 threads with the adaptive lanes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/experiments/ws_check.py default2017 dense 2>&1 | tail -40
timeout 600 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/experiments/seam_b1_ensemble.py
