#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 800 python tools/experiments/lanes_diag2.py 2>&1 | tail -30
