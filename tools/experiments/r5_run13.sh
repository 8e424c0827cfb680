#!/bin/bash
# two scorers on two threads: is it the runtime's handling of scratch (private segment) memory across hardware queues?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=150
for env in "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_NO_SCRATCH_RECLAIM=1" "HSA_NO_SCRATCH_THREAD_LIMITER=1" "HSA_SCRATCH_SINGLE_LIMIT=0" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
