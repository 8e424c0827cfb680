#!/bin/bash
# round 6: lanes for calls of up to 64 poses -- tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 2400 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_parity.py tests/test_host_adapter.py tests/test_gpu_gradient.py tests/test_gpu_cnn_refine.py tests/test_gpu_h2.py -m gpu -x -q 2>&1 | tail -3
python tools/experiments/lanes_max_b.py | tail -9
