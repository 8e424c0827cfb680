#!/bin/bash
# round 5, GPU call 66: after the device guard around the lane streams' creation: the ensemble test
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 100 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q -k "ensemble" 2>&1 | tail -2
