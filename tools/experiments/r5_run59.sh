#!/bin/bash
# round 5, GPU call 59: how the lanes are enqueued (A/B in one process)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
