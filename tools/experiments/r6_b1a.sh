#!/bin/bash
# round 6: the failing device-output concurrency test's message; B = 1 default ensemble with the groups voxelized side by side
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
timeout 300 python tools/experiments/seam_b1_ensemble.py
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_CALL_LOCK=1
timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_NO_LANES=1
