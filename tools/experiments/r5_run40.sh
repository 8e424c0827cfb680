#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 120 tools/microbench/sload_vs_ldsdma 8 2>&1 | tail -8
