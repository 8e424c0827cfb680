#!/bin/bash
# round 6: small synchronous calls hand their poses over in pinned host memory -- tests, seam latency with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrency.py tests/test_host_adapter.py tests/test_gpu_h2.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_LIG_COPY=1
timeout 300 python tools/experiments/seam_b1_ensemble.py
