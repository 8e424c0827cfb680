#!/bin/bash
# round 5, GPU call 48: the device CNN Monte-Carlo chains under --accurate_line_search / --simple_ascent (VERDICT r4 missing #7)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_cnn_refine.py -m gpu -x -q -k "metropolis" 2>&1 | tail -15
