#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=300
for env in "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=0" "HIP_LAUNCH_BLOCKING=1" "MI_VOX_DBG=8"; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
