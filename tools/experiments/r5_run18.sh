#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/experiments/alternate_diag.py 2>&1 | tail -5
