#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=400 MI_GNINA_NO_CALL_LOCK=1
echo "== hit records through vector loads"
timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
