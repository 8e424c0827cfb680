#!/bin/bash
# round 6: gradient calls with single-launch zeroing (gradient maxima, atom-gradient accumulators) -- tests, B = 1 latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_gradient.py tests/test_gpu_cnn_refine.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -3
python tools/experiments/b1_grad_timeline.py default2017
python tools/experiments/b1_grad_timeline.py ensemble
python tools/experiments/b1_grad_timeline.py dense
