#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/experiments/seam_b1_breakdown.py default2017 crossdock_default2018_KD_4 dense_1_3 2>&1 | grep -v amdgpu.ids | tail -12
