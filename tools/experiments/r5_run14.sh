#!/bin/bash
# two scorers on two threads: activations in uncached device memory (is it a stale line in an XCD's L2?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=200
for env in "MI_DEVBUF_UNCACHED=1" ""; do
  echo "== [$env]"
  env $env timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
