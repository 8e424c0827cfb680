#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 800 python tools/experiments/concurrency_diag.py 2>&1 | tail -30
