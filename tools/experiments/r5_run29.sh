#!/bin/bash
# two scorers on two threads with NO LDS-DMA kernel anywhere (the fp32-MFMA program: MI_GNINA_CONV_PATH=f32)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export DIAG_CALLS=400
cat > /tmp/nolock.py <<'PY'
PY
for env in "MI_GNINA_CONV_PATH=f32" ""; do
  echo "== [$env] (per-device lock bypassed: MI_GNINA_NO_CALL_LOCK=1)"
  env $env MI_GNINA_NO_CALL_LOCK=1 timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,dense_1_3_PT_KD_3 dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -4
done
