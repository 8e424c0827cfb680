#!/bin/bash
# two scorers on two threads: which switch makes the deviations go away
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=150
for env in "" "MI_GNINA_NO_DENSE_SPLIT=1" "MI_GNINA_CONV_PATH=0" "MI_GNINA_NO_LAT=1" "MI_GNINA_D16_NP=1" "MI_GNINA_H2_NO_SPLIT_TENSORS=1" "MI_GNINA_H2_WLDS=0" "AMD_SERIALIZE_KERNEL=3" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
