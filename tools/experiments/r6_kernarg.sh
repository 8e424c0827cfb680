#!/bin/bash
# round 6: where the kernel arguments live (HIP_FORCE_DEV_KERNARG) against the per-pose call's latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for V in 0 1; do
echo "== HIP_FORCE_DEV_KERNARG=$V"
HIP_FORCE_DEV_KERNARG=$V timeout 100 tools/microbench/shader_clock_under_load 2>&1 | tail -1
HIP_FORCE_DEV_KERNARG=$V timeout 300 python tools/experiments/seam_b1_ensemble.py
done
echo "== unset"
timeout 300 python tools/experiments/seam_b1_ensemble.py
AMD_LOG_LEVEL=0 python - <<'PY'
import os
print("env:", {k: v for k, v in os.environ.items() if "KERNARG" in k or "HIP_" in k or "HSA_" in k or "GPU_" in k})
PY
