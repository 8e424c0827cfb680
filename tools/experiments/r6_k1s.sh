#!/bin/bash
# round 6: chunk groups in the 1x1x1 transitions of per-pose calls -- bits, seam latency with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_dense_split.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_K1S_DBG=128
timeout 300 python tools/experiments/seam_b1_ensemble.py
python tools/experiments/dense_throughput.py | tail -1
