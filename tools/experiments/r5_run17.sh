#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=250
MI_GNINA_H2_NO_SPLIT_TENSORS=1 timeout 600 python tools/experiments/concurrency_diag2.py crossdock_default2018_KD_4 dense_1_3 2>&1 | tail -16
MI_GNINA_H2_NO_SPLIT_TENSORS=1 timeout 600 python tools/experiments/concurrency_diag2.py dense_1_3_PT_KD_3 dense_1_3 2>&1 | tail -16
