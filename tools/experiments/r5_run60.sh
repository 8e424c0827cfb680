#!/bin/bash
# round 5, GPU call 60: the global max pool with eight threads per channel: Dense parity, B = 1 latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_custom_model.py -m gpu -x -q -k "goldens or ensemble or custom or given_grids or batch_independence" 2>&1 | tail -3
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
