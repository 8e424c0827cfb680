#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/experiments/concurrency_diag2.py crossdock_default2018_KD_4 dense_1_3 2>&1 | tail -22
timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -22
MI_GNINA_CONV_PATH=0 timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -22
